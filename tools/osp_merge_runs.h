// osp_merge_runs.h -- the merge phase as a MERGE of pre-sorted runs (LDS, gfx950 wave64).
//
// Reference: deduplicateCOO, SimSpGEMM.cpp:519-535 -- sort all partial products by (row, col),
// sum equal keys left to right, keep zeros.  The reference sorts from scratch; here the structure
// the multiply phase leaves behind is used instead: a row's partial products are `deg_A(row)`
// chunks (one per A[row,k], ascending k), and every chunk is a scaled copy of a B row, so it is
// ALREADY sorted by column (the reference's own merge-tree idea, merge2way/mergeHardware
// SimSpGEMM.cpp:306-441, is the same observation).  A tile of consecutive rows is staged in LDS
// and its chunks are merged pairwise, level by level: every element finds its rank in the sibling
// run by binary search (left run wins ties, so equal columns stay in ascending-k order -- the
// oracle's stable-sort order), log2(chunks per row) levels instead of 4 radix passes.
//
// Tiles are handed out in output order by a ticket counter; a decoupled look-back over
// `tile_status` gives every tile its offset in the final CSR, so merged rows are written once.
#pragma once
#include "osp_kernels.h"

namespace osp {

template <class T> struct RunCap;
template <> struct RunCap<float> { static constexpr int value = 3584; };
template <> struct RunCap<double> { static constexpr int value = 3072; };
constexpr int kRunsThreads = 512;

template <class T, int NT>
struct RunsSmem {
    static constexpr int CAP = RunCap<T>::value;
    uint32_t key[2][CAP];   // column
    uint32_t pay[2][CAP];   // staging position (12) | chunk ordinal in its row (12) | local row (8)
    T val[CAP];
    uint16_t aux[CAP + 2];  // chunk starts while merging, output slots afterwards
    uint16_t cfirst[kTileMaxRows + 2];
    uint32_t rowo[kTileMaxRows + 1];
    uint32_t scratch[NT / kWave + 1];
    uint32_t tile, maxdeg;
    uint64_t excl;
};

// ABL (tools/bench_merge.hip only): 1 = skip the merge levels, 2 = no look-back, 4 = no ticket.
template <class T, int NT, int ABL = 0>
__global__ __launch_bounds__(NT) void merge_runs_kernel(
    const uint32_t *__restrict__ tile_rows, uint32_t ntiles, uint64_t r_end,
    const uint64_t *__restrict__ row_off, uint64_t base, const uint32_t *__restrict__ arow,
    const uint64_t *__restrict__ chunk_start, const Part<T> *__restrict__ stage,
    const uint32_t *__restrict__ heavy_nnz, uint64_t *tile_status,
    uint32_t *ticket, const uint64_t *__restrict__ out_base_p, int64_t *__restrict__ c_rowptr,
    uint32_t *__restrict__ c_col, T *__restrict__ c_val, uint64_t *__restrict__ out_end_p) {
    __shared__ RunsSmem<T, NT> sm;
    constexpr int CAP = RunCap<T>::value;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    if (tid == 0) {
        sm.tile = (ABL & 4) ? blockIdx.x : atomicAdd(ticket, 1u);
        sm.maxdeg = 0;
    }
    __syncthreads();
    const uint32_t t = sm.tile;
    if (t >= ntiles) return;
    const uint64_t out_base = *out_base_p;
    const uint64_t ra = tile_rows[t];
    const uint64_t rb = (t + 1 < ntiles) ? (uint64_t)tile_rows[t + 1] : r_end;
    const uint32_t nr = (uint32_t)(rb - ra);
    const uint64_t s = row_off[ra] - base;
    const uint32_t n = (uint32_t)min(row_off[rb] - base - s, (uint64_t)CAP + 1);
    if (n > (uint32_t)CAP) {
        // a single long row, already reduced in place by the global-sort path: only take part in the
        // offset chain; heavy_copy_kernel moves its entries once c_rowptr is known
        if (w == 0) {
            const uint64_t total = heavy_nnz[ra];
            const uint64_t excl = (ABL & 2) ? (uint64_t)t * CAP : lookback_prefix(tile_status, t, total);
            if (lane == 0) {
                c_rowptr[ra] = (int64_t)(out_base + excl);
                if (t + 1 == ntiles) { c_rowptr[r_end] = (int64_t)(out_base + excl + total); *out_end_p = out_base + excl + total; }
            }
        }
        return;
    }
    // ---- tile metadata: chunk starts, first chunk of every row, row starts ----
    const uint32_t c0 = arow[ra];
    const uint32_t nc = arow[rb] - c0;  // non-empty chunks in the tile (<= n)
    for (uint32_t c = tid; c <= nc; c += NT) sm.aux[c] = (uint16_t)(chunk_start[c0 + c] - base - s);
    uint32_t mydeg = 0;
    for (uint32_t r = tid; r <= nr; r += NT) {
        const uint32_t cf = arow[ra + r] - c0;
        sm.cfirst[r] = (uint16_t)cf;
        sm.rowo[r] = (uint32_t)(row_off[ra + r] - base - s);
        if (r < nr) mydeg = max(mydeg, arow[ra + r + 1] - c0 - cf);
    }
    if (mydeg) atomicMax(&sm.maxdeg, mydeg);
    __syncthreads();
    // ---- stage the tile ----
    for (uint32_t p = tid; p < n; p += NT) {
        uint32_t lo = 0, hi = nc;  // chunk of p: last c < nc with aux[c] <= p
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sm.aux[mid] <= p) lo = mid; else hi = mid; }
        const uint32_t c = lo;
        lo = 0; hi = nr;           // row of c: last r < nr with cfirst[r] <= c
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sm.cfirst[mid] <= c) lo = mid; else hi = mid; }
        const Part<T> pp = stage[s + p];
        sm.key[0][p] = pp.col;
        sm.pay[0][p] = p | ((c - sm.cfirst[lo]) << 12) | (lo << 24);
        sm.val[p] = pp.val;
    }
    __syncthreads();
    int levels = 0;
    if (!(ABL & 1)) while ((1u << levels) < sm.maxdeg) levels++;
    int cur = 0;
    // ---- merge levels: runs of 2^j chunks -> runs of 2^(j+1) chunks, inside every row ----
    for (int j = 0; j < levels; j++) {
        const uint32_t span = 1u << j;
        for (uint32_t p = tid; p < n; p += NT) {
            const uint32_t k = sm.key[cur][p], py = sm.pay[cur][p];
            const uint32_t ord = (py >> 12) & 0xfffu, r = py >> 24;
            const uint32_t cf = sm.cfirst[r], ce = sm.cfirst[r + 1];
            const uint32_t g = ord >> j, f = cf + (g << j);
            const uint32_t own = sm.aux[f];
            const bool left_sib = g & 1u;
            uint32_t lo, hi, mstart;
            if (left_sib) { lo = sm.aux[f - span]; hi = own; mstart = lo; }
            else { const uint32_t sf = min(f + span, ce); lo = sm.aux[sf]; hi = sm.aux[min(sf + span, ce)]; mstart = own; }
            // rank in the sibling run: left sibling wins ties (its chunks have smaller k)
            uint32_t a = lo, b = hi;
            while (a < b) {
                const uint32_t m = (a + b) >> 1;
                const uint32_t kv = sm.key[cur][m];
                const bool go = left_sib ? (kv <= k) : (kv < k);
                if (go) a = m + 1; else b = m;
            }
            const uint32_t np = mstart + (p - own) + (a - lo);
            sm.key[cur ^ 1][np] = k;
            sm.pay[cur ^ 1][np] = py;
        }
        cur ^= 1;
        __syncthreads();
    }
    // ---- head flags + exclusive scan (blocked: thread owns IPT consecutive sorted entries) ----
    constexpr int IPT = (CAP + NT - 1) / NT;
    const uint32_t ib = tid * IPT;
    uint32_t heads = 0, hmask = 0;
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint32_t i = ib + q;
        if (i < n) {
            const bool h = (i == 0) || (sm.key[cur][i] != sm.key[cur][i - 1]) ||
                           ((sm.pay[cur][i] >> 24) != (sm.pay[cur][i - 1] >> 24));
            hmask |= (h ? 1u : 0u) << q;
            heads += h;
        }
    }
    uint32_t total;
    uint32_t ex = block_excl_scan<uint32_t, NT>(heads, sm.scratch, &total);
    // the tile's unique count is known: start the look-back now, it overlaps the run sums below
    if (w == 0) {
        const uint64_t excl = (ABL & 2) ? (uint64_t)t * CAP : lookback_prefix(tile_status, t, total);
        if (lane == 0) sm.excl = excl;
    }
    uint32_t oslot[IPT];
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint32_t i = ib + q;
        oslot[q] = ex;
        if (i < n) {
            sm.aux[i] = (uint16_t)ex;  // output slot of the run that starts at/behind i
            ex += (hmask >> q) & 1u;
        }
    }
    if (tid == 0) sm.aux[n] = (uint16_t)total;
    // each head sums its run in staging order (= ascending k)
    T acc[IPT];
    uint32_t ocol[IPT];
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint32_t i = ib + q;
        acc[q] = 0; ocol[q] = 0;
        if (i < n && ((hmask >> q) & 1u)) {
            const uint32_t k = sm.key[cur][i], rr = sm.pay[cur][i] >> 24;
            T a = sm.val[sm.pay[cur][i] & 0xfffu];
            for (uint32_t u = i + 1; u < n && sm.key[cur][u] == k && (sm.pay[cur][u] >> 24) == rr; u++)
                a += sm.val[sm.pay[cur][u] & 0xfffu];
            acc[q] = a;
            ocol[q] = k;
        }
    }
    __syncthreads();  // all gathers from val[] / key[cur] done; look-back result is in sm.excl
    // compact into LDS (val[] and the idle key buffer), then stream out with consecutive lanes on
    // consecutive addresses
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint32_t i = ib + q;
        if (i < n && ((hmask >> q) & 1u)) { sm.key[cur ^ 1][oslot[q]] = ocol[q]; sm.val[oslot[q]] = acc[q]; }
    }
    __syncthreads();
    const uint64_t obase = out_base + sm.excl;
    for (uint32_t o = tid; o < total; o += NT) { c_col[obase + o] = sm.key[cur ^ 1][o]; c_val[obase + o] = sm.val[o]; }
    // rows keep their index span through the merge
    for (uint32_t r = tid; r < nr; r += NT) c_rowptr[ra + r] = (int64_t)(obase + sm.aux[sm.rowo[r]]);
    if (t + 1 == ntiles && tid == 0) { c_rowptr[r_end] = (int64_t)(obase + total); *out_end_p = obase + total; }
}

// ---- chunk list: non-empty chunks in (row, k) order ----------------------------------------------
// cand = every candidate chunk in row-major order with its length; keep the non-empty ones.
template <class LenF>
struct NonEmptyFlag {
    LenF len;
    __device__ uint32_t operator()(uint64_t c) const { return len(c) > 0 ? 1u : 0u; }
};
template <class LenF>
__global__ void chunk_compact_kernel(LenF len, const uint64_t *offs, const uint32_t *flscan, uint64_t ncand,
                                     uint64_t *chunk_start) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncand) {
        if (len(c) > 0) chunk_start[flscan[c]] = offs[c];
    } else if (c == ncand) {
        chunk_start[flscan[ncand]] = offs[ncand];  // sentinel = total number of partial products
    }
}

}  // namespace osp
