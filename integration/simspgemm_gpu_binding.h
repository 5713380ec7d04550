// simspgemm_gpu_binding.h -- the binding a maintainer of anneouyang/OuterSPACE would add to simulator/SimSpGEMM.cpp to
// run the numeric SpGEMM on an MI355X (INTEGRATION.md section 2 quotes this file).
//
// Include AFTER the reference's "common.h" (it uses CSRMatrix, CSRElement, COOMatrix, COOElement, index_t, value_t:
// common.h:7-49) and link with -louterspace_spgemm.  Replaces, in one call,
//     deduplicateCOO(concat(cscMulcsr(csc, csr)))          SimSpGEMM.cpp:265-281 + :519-535
// The operands go to the library exactly as the reference holds them -- std::vector<size_t> pos and the packed
// std::vector<CSRElement{idx,val}> data -- through osp_spgemm_csc_csr_aos: no conversion, no copy on the host.
#pragma once
#include <cassert>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "outerspace_spgemm.h"

inline COOMatrix cscMulcsrMergedGPU(const CSRMatrix &csc, const CSRMatrix &csr, size_t NRowA, size_t NColB) {
    static_assert(sizeof(size_t) == sizeof(uint64_t), "CSRMatrix::pos is handed over as 64-bit offsets");
    static_assert(sizeof(CSRElement) == sizeof(index_t) + sizeof(value_t), "CSRElement is packed (common.h:10-16)");
    static_assert(sizeof(index_t) == 4 && (sizeof(value_t) == 4 || sizeof(value_t) == 8), "u32 indices, f32 or f64 values");
    assert(csc.pos.size() == csr.pos.size());  // SimSpGEMM.cpp:267
    const size_t K = csc.pos.size() - 1;
    const osp_dtype_t dt = sizeof(value_t) == 4 ? OSP_F32 : OSP_F64;

    static osp_context_t ctx = nullptr;  // one GPU context for the process, like the reference's single thread
    if (!ctx && osp_context_create(0, &ctx)) throw std::runtime_error(osp_last_error_string());
    osp_result_t res = nullptr;
    const int st = osp_spgemm_csc_csr_aos(ctx, dt, NRowA, K, NColB, reinterpret_cast<const uint64_t *>(csc.pos.data()),
                                          csc.data.data(), reinterpret_cast<const uint64_t *>(csr.pos.data()), csr.data.data(),
                                          OSP_HOST, nullptr, &res);
    if (st == OSP_ERR_DUPLICATE) throw(233);  // what dupcheck throws, SimSpGEMM.cpp:49
    if (st) throw std::runtime_error(osp_last_error_string());

    osp_result_info_t info;
    osp_result_info(res, &info);  // info.partials == mulflops_ref (SimSpGEMM.cpp:884-891)
    std::vector<int64_t> rowptr(NRowA + 1);
    std::vector<uint32_t> col(info.nnz_c);
    std::vector<value_t> val(info.nnz_c);
    const int st2 = osp_result_copy_csr(res, rowptr.data(), col.data(), val.data(), OSP_HOST);
    osp_result_destroy(res);
    if (st2) throw std::runtime_error(osp_last_error_string());

    COOMatrix out;  // sorted by (row, col), equal coordinates summed: what deduplicateCOO returns
    out.reserve(info.nnz_c);
    for (size_t r = 0; r < NRowA; r++)
        for (int64_t i = rowptr[r]; i < rowptr[r + 1]; i++) out.push_back(COOElement{index_t(r), col[i], val[i]});
    return out;
}
