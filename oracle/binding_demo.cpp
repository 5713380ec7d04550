// binding_demo.cpp -- TEST INFRASTRUCTURE ONLY: the reference's own code calling the GPU library through the binding
// of integration/simspgemm_gpu_binding.h.  Built by oracle/Makefile (build container only, where /root/reference
// exists) into oracle/_ref/binding_demo; tests/test_gpu_parity.py runs it on the GPU box.
//
// It performs main()'s data flow with the REFERENCE's functions (SimSpGEMM.cpp: readcoo :55-100, the transpose of the
// second operand :852-856, coo2csr<true>/coo2csr :878-879, mulflops_ref :884-891), then forms the product twice --
// cscMulcsr + deduplicateCOO on the CPU (:265-281, :519-535) and cscMulcsrMergedGPU on the MI355X -- and compares them:
// coordinates identical, values within 1e-6 relative (f64) / 1e-5 (f32).  Exit code 0 = match.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <fstream>
#include <sstream>
#include <vector>
#include <algorithm>
#include <chrono>
#include <deque>
#include <queue>
#include <set>
#include <map>
#include <cassert>
#include <cmath>
#include <list>
#include <unordered_map>

#ifdef OSP_REF_F64
#define float double
#endif
#include "simulator/common.h"
#include "simulator/SimCache.h"
#define main static __attribute__((unused)) osp_ref_unused_main
#include OSP_REF_TU
#undef main
#ifdef OSP_REF_F64
#undef float
#endif

#include "simspgemm_gpu_binding.h"

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s A.mtx B.mtx\n", argv[0]); return 2; }
    size_t NRow[2], NCol[2];
    COOMatrix coo[2];
    for (size_t i = 0; i < 2; i++) {
        std::ifstream fin(argv[1 + i]);
        if (!fin) { fprintf(stderr, "cannot open %s\n", argv[1 + i]); return 2; }
        coo[i] = readcoo(fin, NRow[i], NCol[i], false);
    }
    std::swap(NRow[1], NCol[1]);
    for (auto &&e : coo[1]) std::swap(e.row, e.col);
    CSRMatrix csc = coo2csr<true>(coo[0], NCol[0]);
    CSRMatrix csr = coo2csr(coo[1], NRow[1]);
    size_t mulflops = 0;
    for (size_t i = 0; i + 1 < csr.pos.size(); i++) mulflops += (csc.pos[i + 1] - csc.pos[i]) * (csr.pos[i + 1] - csr.pos[i]);
    printf("mul flops ref = %zu\n", mulflops);

    COOMatrix all;
    for (auto &p : cscMulcsr(csc, csr)) all.insert(all.end(), p.begin(), p.end());
    const COOMatrix want = all.empty() ? COOMatrix() : deduplicateCOO(std::move(all));
    COOMatrix got;
    try {
        got = cscMulcsrMergedGPU(csc, csr, NRow[0], NCol[1]);
    } catch (const std::exception &e) {
        fprintf(stderr, "GPU product failed: %s\n", e.what());
        return 3;
    } catch (int code) {  // 233: duplicate coordinate, as the reference throws it
        fprintf(stderr, "GPU product threw %d\n", code);
        return code;
    }

    if (got.size() != want.size()) { printf("MISMATCH: %zu entries on the GPU, %zu from the reference\n", got.size(), want.size()); return 1; }
    const double tol = sizeof(value_t) == 4 ? 1e-5 : 1e-6;
    // Relative to the entry, or -- for entries that are the small remainder of terms of both signs (the MLP layers) --
    // to the largest entry: the two sides add equal keys in different orders (the reference's std::sort is unstable).
    double worst = 0, amax = 1e-300;
    bool mixed = false;
    for (const auto &e : want) { amax = std::max(amax, std::fabs((double)e.val)); mixed = mixed || e.val < 0; }
    for (size_t i = 0; i < want.size(); i++) {
        if (got[i].row != want[i].row || got[i].col != want[i].col) { printf("MISMATCH: coordinate %zu differs\n", i); return 1; }
        const double d = std::fabs((double)got[i].val - (double)want[i].val);
        const double s = mixed ? amax : std::max(std::fabs((double)want[i].val), 1e-300);
        worst = std::max(worst, d / s);
    }
    if (worst > tol) { printf("MISMATCH: values differ by %.3g relative\n", worst); return 1; }
    printf("MATCH: %zu entries, coordinates identical, max relative difference %.3g (value_t = %zu bytes)\n", want.size(), worst, sizeof(value_t));
    return 0;
}
