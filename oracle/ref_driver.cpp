// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the REFERENCE's own translation unit simulator/SimSpGEMM.cpp, in place
// under /root/reference (found through -I, see oracle/Makefile; never copied), and
// exposes its live functions through a small C ABI so that
//   * tests/golden/make_golden.py can generate golden vectors, and
//   * bench.py can time the reference algorithm as cpu_baseline.kind="reference".
// Output goes to oracle/_ref/ only (git-ignored; it still travels to the GPU box).
//
// How the reference TU is taken in without touching it:
//   * `main` is renamed to an unused static function, so its one unresolvable call
//     (simulateOuterSPACE, defined in SimOuterSPACE.cpp which needs the absent
//     ramulator) is discarded with it -- no stand-in is written for anything.
//   * the f64 build (-DOSP_REF_F64) compiles the same TU with `float` spelled
//     `double`, which turns `typedef float value_t` (common.h:8) into double; the
//     system headers the TU uses are included first so the macro cannot reach them.
//   * deduplicateCOO (SimSpGEMM.cpp:519-535) sits inside `#if 0` and cannot be
//     compiled in place; ref_merge() below performs its three statements with the
//     reference's own COOElement::operator< (common.h:29-32) and std::sort.
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

#ifdef OSP_REF_F64
#define float double
#endif
#define main static __attribute__((unused)) osp_ref_unused_main
#include "simulator/SimSpGEMM.cpp"
#undef main
#ifdef OSP_REF_F64
#undef float
#endif

namespace {

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// deduplicateCOO, SimSpGEMM.cpp:519-535 (guarded for the empty input the original
// would dereference).
COOMatrix ref_merge(COOMatrix coo) {
    COOMatrix result;
    if (coo.empty()) return result;
    std::sort(coo.begin(), coo.end());
    result.push_back(coo.front());
    for (size_t i = 1; i < coo.size(); i++) {
        if (coo[i].row != coo[i - 1].row || coo[i].col != coo[i - 1].col)
            result.push_back(coo[i]);
        else
            result.back().val += coo[i].val;
    }
    return result;
}

int export_coo(const COOMatrix &c, uint64_t *n, uint32_t **rows, uint32_t **cols, value_t **vals) {
    *n = c.size();
    size_t m = c.size() ? c.size() : 1;
    *rows = (uint32_t *)malloc(m * sizeof(uint32_t));
    *cols = (uint32_t *)malloc(m * sizeof(uint32_t));
    *vals = (value_t *)malloc(m * sizeof(value_t));
    for (size_t i = 0; i < c.size(); i++) {
        (*rows)[i] = c[i].row;
        (*cols)[i] = c[i].col;
        (*vals)[i] = c[i].val;
    }
    return 0;
}

}  // namespace

extern "C" {

int osp_ref_value_size() { return (int)sizeof(value_t); }
void osp_ref_free(void *p) { free(p); }

// readcoo, SimSpGEMM.cpp:55-100.
int osp_ref_readcoo(const char *path, int sym, uint64_t *nrow, uint64_t *ncol, uint64_t *nnz,
                    uint32_t **rows, uint32_t **cols, value_t **vals) {
    std::ifstream fin(path);
    if (!fin) return 4;
    size_t NRow, NCol;
    COOMatrix coo = readcoo(fin, NRow, NCol, sym != 0);
    *nrow = NRow;
    *ncol = NCol;
    return export_coo(coo, nnz, rows, cols, vals);
}

// coo2csr<transpose>, SimSpGEMM.cpp:102-152.  pos must hold nseg+1 entries.
// Returns 233 when the reference throws 233 (duplicate coordinate, :49).
int osp_ref_coo2csr(int transpose, uint64_t nseg, uint64_t nnz, const uint32_t *rows,
                    const uint32_t *cols, const value_t *vals, int64_t *pos, uint32_t *idx,
                    value_t *out_val) {
    COOMatrix coo(nnz);
    for (size_t i = 0; i < nnz; i++) coo[i] = COOElement{rows[i], cols[i], vals[i]};
    try {
        CSRMatrix m = transpose ? coo2csr<true>(coo, nseg) : coo2csr<false>(coo, nseg);
        for (size_t i = 0; i <= nseg; i++) pos[i] = (int64_t)m.pos[i];
        for (size_t i = 0; i < nnz; i++) {
            idx[i] = m.data[i].idx;
            out_val[i] = m.data[i].val;
        }
    } catch (int e) {
        return e;
    }
    return 0;
}

// The numeric path of SURVEY.md section 3(b) on already-compressed operands, for
// the k-slab [k0,k1):  cscMulcsr (:265-281) -> concat -> sort+sum (:519-535).
// Result is COO sorted by (row,col).  secs = {multiply+concat, merge} seconds.
int osp_ref_spgemm_csx(uint64_t K, uint64_t k0, uint64_t k1, const int64_t *a_pos,
                       const uint32_t *a_idx, const value_t *a_val, const int64_t *b_pos,
                       const uint32_t *b_idx, const value_t *b_val, uint64_t *nnzc,
                       uint64_t *partials, uint32_t **rows, uint32_t **cols, value_t **vals,
                       double *secs) {
    if (k1 > K || k0 > k1) return 2;
    CSRMatrix csc, csr;
    size_t nk = k1 - k0;
    csc.pos.resize(nk + 1);
    csr.pos.resize(nk + 1);
    for (size_t k = 0; k <= nk; k++) {
        csc.pos[k] = (size_t)(a_pos[k0 + k] - a_pos[k0]);
        csr.pos[k] = (size_t)(b_pos[k0 + k] - b_pos[k0]);
    }
    csc.data.resize(csc.pos[nk]);
    csr.data.resize(csr.pos[nk]);
    for (size_t i = 0; i < csc.data.size(); i++)
        csc.data[i] = CSRElement{a_idx[a_pos[k0] + i], a_val[a_pos[k0] + i]};
    for (size_t i = 0; i < csr.data.size(); i++)
        csr.data[i] = CSRElement{b_idx[b_pos[k0] + i], b_val[b_pos[k0] + i]};

    double t0 = now();
    std::vector<COOMatrix> parts = cscMulcsr(csc, csr);
    COOMatrix all;
    size_t P = 0;
    for (auto &p : parts) P += p.size();
    all.reserve(P);
    for (auto &p : parts) all.insert(all.end(), p.begin(), p.end());
    std::vector<COOMatrix>().swap(parts);
    double t1 = now();
    COOMatrix c = ref_merge(std::move(all));
    double t2 = now();
    if (partials) *partials = P;
    if (secs) {
        secs[0] = t1 - t0;
        secs[1] = t2 - t1;
    }
    if (!rows) {  // timing-only call
        *nnzc = c.size();
        return 0;
    }
    return export_coo(c, nnzc, rows, cols, vals);
}

// The CLI data flow of main(), SimSpGEMM.cpp:819-891, from two .mtx paths:
// readcoo x2 (:844-850), transpose the second operand (:852-856, optional here),
// coo2csr<true>(A, NCol_A) / coo2csr(B', NRow_B') (:878-879), the inner-dimension
// assert (:882, returned as code 5), mulflops_ref (:884-891), then the numeric path.
int osp_ref_spgemm_mtx(const char *path_a, const char *path_b, int transpose_b, uint64_t *M,
                       uint64_t *N, uint64_t *nnzc, uint64_t *partials, uint32_t **rows,
                       uint32_t **cols, value_t **vals) {
    size_t NRow[2], NCol[2];
    COOMatrix coo[2];
    const char *fn[2] = {path_a, path_b};
    for (size_t i = 0; i < 2; i++) {
        std::ifstream fin(fn[i]);
        if (!fin) return 4;
        coo[i] = readcoo(fin, NRow[i], NCol[i], false);
    }
    if (transpose_b) {
        std::swap(NRow[1], NCol[1]);
        for (auto &&e : coo[1]) std::swap(e.row, e.col);
    }
    try {
        CSRMatrix csc = coo2csr<true>(coo[0], NCol[0]);
        CSRMatrix csr = coo2csr(coo[1], NRow[1]);
        if (csr.pos.size() != csc.pos.size()) return 5;
        size_t mulflops = 0;
        for (size_t i = 0; i + 1 < csr.pos.size(); i++)
            mulflops += (csc.pos[i + 1] - csc.pos[i]) * (csr.pos[i + 1] - csr.pos[i]);
        std::vector<COOMatrix> parts = cscMulcsr(csc, csr);
        COOMatrix all;
        for (auto &p : parts) all.insert(all.end(), p.begin(), p.end());
        COOMatrix c = ref_merge(std::move(all));
        *M = NRow[0];
        *N = NCol[1];
        *partials = mulflops;
        return export_coo(c, nnzc, rows, cols, vals);
    } catch (int e) {
        return e;
    }
}

}  // extern "C"
