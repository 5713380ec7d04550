// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Compiles the REFERENCE's own translation unit simulator/SimSpGEMM.cpp and exposes its
// functions through a small C ABI so that
//   * tests/golden/make_golden.py can generate golden vectors,
//   * tests/test_oracle_golden.py can cross-check the reference's three producers and two mergers
//     against each other and against the plain-C restatement, and
//   * bench.py can time the reference algorithm as cpu_baseline.kind="reference".
// Output goes to oracle/_ref/ only (git-ignored; the .so still travels to the GPU box).
//
// How the reference TU is taken in (recipe: oracle/Makefile):
//   * The merge half of the path -- deduplicateCOO (:519-535), merge2way (:306-327), multHardware
//     (:358-409), mergeHardware (:411-441), merge (:445-517) -- sits between `#if 0` (line 304) and
//     `#endif` (line 812).  The Makefile writes the TU with that ONE line changed to `#if 1` into a
//     mktemp directory OUTSIDE the repository (`sed '304s/^#if 0$/#if 1/'`, after checking that line
//     304 is exactly `#if 0`), passes the directory with -I, and deletes it when the compiler returns.
//     Nothing of the reference's text is copied into the tree; OSP_REF_TU names that temporary file.
//   * The enabled block needs SimCache (`SimCache cache(8, 13, 3)`, :340): the reference's own
//     simulator/SimCache.h is included in place (the TU's own include of it is commented out, :18).
//   * `main` is renamed to an unused static function, so its one unresolvable call
//     (simulateOuterSPACE, defined in SimOuterSPACE.cpp which needs the absent ramulator) is
//     discarded with it -- no stand-in is written for anything.
//   * the f64 build (-DOSP_REF_F64) compiles the same TU with `float` spelled `double`, which turns
//     `typedef float value_t` (common.h:8) into double; the system headers the TU uses are included
//     first so the macro cannot reach them.
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

#ifndef OSP_REF_TU
#error "build through oracle/Makefile: OSP_REF_TU names the reference TU with its merge block enabled"
#endif

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

namespace {

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
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

COOMatrix concat(std::vector<COOMatrix> &parts, size_t *P) {
    COOMatrix all;
    size_t n = 0;
    for (auto &p : parts) n += p.size();
    all.reserve(n);
    for (auto &p : parts) all.insert(all.end(), p.begin(), p.end());
    std::vector<COOMatrix>().swap(parts);
    if (P) *P = n;
    return all;
}

// the reference's deduplicateCOO (:519-535) reads coo.front() of an empty input
COOMatrix dedup(COOMatrix all) {
    if (all.empty()) return COOMatrix();
    return deduplicateCOO(std::move(all));
}

// slab [k0,k1) of compressed SoA operands -> the reference's CSRMatrix
void load_slab(uint64_t k0, uint64_t k1, const int64_t *pos, const uint32_t *idx, const value_t *val, CSRMatrix &m) {
    const size_t nk = k1 - k0;
    m.pos.resize(nk + 1);
    for (size_t k = 0; k <= nk; k++) m.pos[k] = (size_t)(pos[k0 + k] - pos[k0]);
    m.data.resize(m.pos[nk]);
    for (size_t i = 0; i < m.data.size(); i++) m.data[i] = CSRElement{idx[pos[k0] + i], val[pos[k0] + i]};
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
// the k-slab [k0,k1):  cscMulcsr (:265-281) -> concat -> deduplicateCOO (:519-535).
// Result is COO sorted by (row,col).  secs = {multiply+concat, merge} seconds.
int osp_ref_spgemm_csx(uint64_t K, uint64_t k0, uint64_t k1, const int64_t *a_pos,
                       const uint32_t *a_idx, const value_t *a_val, const int64_t *b_pos,
                       const uint32_t *b_idx, const value_t *b_val, uint64_t *nnzc,
                       uint64_t *partials, uint32_t **rows, uint32_t **cols, value_t **vals,
                       double *secs) {
    if (k1 > K || k0 > k1) return 2;
    CSRMatrix csc, csr;
    load_slab(k0, k1, a_pos, a_idx, a_val, csc);
    load_slab(k0, k1, b_pos, b_idx, b_val, csr);
    double t0 = now();
    std::vector<COOMatrix> parts = cscMulcsr(csc, csr);
    size_t P = 0;
    COOMatrix all = concat(parts, &P);
    double t1 = now();
    COOMatrix c = dedup(std::move(all));
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

// The same product through the reference's ALTERNATIVE producers and mergers (SURVEY.md 8a rows
// "merge2way/mergeHardware/merge/multHardware" and "csr2compact/compactMulcsr/csc2rawcompact"):
//   variant 0  deduplicateCOO(concat(cscMulcsr(csc, csr)))                        :265-281 + :519-535
//   variant 1  deduplicateCOO(concat(compactMulcsr(csc2rawcompact(csc), csr)))    :221-242, :247-263
//   variant 2  deduplicateCOO(concat(compactMulcsr(csr2compact(A_csr), csr)))     :154-219, :247-263
//   variant 3  merge(csr2compact(A_csr), csr)   -- multHardware + the 6-layer merge2way tree      :306-517
// A_csr = coo2csr<false>(A as COO, M), the reference's own conversion (:102-152).
// Return codes: 0 ok; 2 bad argument; 233 = dupcheck threw (:49; compactMulcsr and mergeHardware call it);
// 6 = the reference's merge() would fail its own assert(tmp.size() <= mergeK) (:483): when A's longest row has
// more than MAX_MERGE_K = 64 non-zeros and that count is a multiple of 63, mergeK starts at 0 (:457).
int osp_ref_spgemm_variant(int variant, uint64_t M, uint64_t K, const int64_t *a_pos, const uint32_t *a_idx,
                           const value_t *a_val, const int64_t *b_pos, const uint32_t *b_idx,
                           const value_t *b_val, uint64_t *nnzc, uint64_t *partials, uint32_t **rows,
                           uint32_t **cols, value_t **vals) {
    if (variant < 0 || variant > 3) return 2;
    CSRMatrix csc, csr;
    load_slab(0, K, a_pos, a_idx, a_val, csc);
    load_slab(0, K, b_pos, b_idx, b_val, csr);
    try {
        COOMatrix c;
        size_t P = 0;
        if (variant == 0) {
            std::vector<COOMatrix> parts = cscMulcsr(csc, csr);
            c = dedup(concat(parts, &P));
        } else if (variant == 1) {
            std::vector<COOMatrix> parts = compactMulcsr(csc2rawcompact(csc), csr);
            c = dedup(concat(parts, &P));
        } else {
            // A in row-major form through the reference's own conversion
            const CompactCOOMatrix raw = csc2rawcompact(csc);
            const CSRMatrix a_csr = coo2csr<false>(raw.data, (size_t)M);
            const CompactCOOMatrix compact = csr2compact(a_csr);
            for (size_t k = 0; k < K; k++) P += (csc.pos[k + 1] - csc.pos[k]) * (csr.pos[k + 1] - csr.pos[k]);
            if (variant == 2) {
                std::vector<COOMatrix> parts = compactMulcsr(compact, csr);
                c = dedup(concat(parts, nullptr));
            } else {
                const size_t ways = compact.pos.empty() ? 0 : compact.pos.size() - 1;
                if (ways > MAX_MERGE_K && ways % (MAX_MERGE_K - 1) == 0) return 6;
                c = merge(compact, csr);
                simRowOrder.clear();  // the reference's global access log (:344) only grows
                simRowOrder.shrink_to_fit();
            }
        }
        if (partials) *partials = P;
        return export_coo(c, nnzc, rows, cols, vals);
    } catch (int e) {
        return e;
    }
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
        COOMatrix c = dedup(concat(parts, nullptr));
        *M = NRow[0];
        *N = NCol[1];
        *partials = mulflops;
        return export_coo(c, nnzc, rows, cols, vals);
    } catch (int e) {
        return e;
    }
}

}  // extern "C"
