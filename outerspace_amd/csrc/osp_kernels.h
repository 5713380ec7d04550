// osp_kernels.h -- HIP kernels of the outer-product SpGEMM for gfx950 (MI355X, wave64).
//
// Pipeline (host orchestration in osp_api.hip):
//   symbolic   every non-zero A[i,k] owns one CHUNK of nnz(B[k,:]) partial products; chunks of one
//              output row are laid out contiguously, ordered by k, so a row's partial products are
//              one contiguous span of the staging buffer and the row id is implicit.
//   multiply   cscMulcsr (SimSpGEMM.cpp:265-281): column k of A x row k of B, written chunk by
//              chunk.  The product space is flattened and cut into equal slices, one per wave.
//   merge      deduplicateCOO (SimSpGEMM.cpp:519-535): per tile of consecutive rows, a stable LSD
//              radix sort on (row, col) in LDS, equal keys summed in staging order (= ascending
//              k, the order the oracle's stable sort yields), result written back in place.
//              Rows too long for LDS take the global-sort path.
//   compact    per-row results -> final CSR at exact offsets.
#pragma once
#include "osp_prims.h"

namespace osp {

// ---- tunables --------------------------------------------------------------------------------
// partial products one LDS merge tile holds: sized so that two workgroups fit the CU's 160 KiB LDS
template <class T> struct TileCap;
template <> struct TileCap<float> { static constexpr int value = 4096; };
template <> struct TileCap<double> { static constexpr int value = 3072; };
constexpr int kTileMaxRows = 256;  // rows per tile (bounds the row bits of the sort key)
constexpr int kMergeThreads = 256;
constexpr int kMulThreads = 256;
constexpr int kMulPerWave = 2048;  // partial products per wave slice
constexpr int kMulPerBlock = kMulPerWave * (kMulThreads / kWave);

// error flag bits written by validate kernels
constexpr uint32_t kFlagRange = 1u, kFlagUnsorted = 2u, kFlagDuplicate = 4u, kFlagPtr = 8u;

// ---- small helpers -----------------------------------------------------------------------------
// first index in [lo,hi) with a[idx] > x   (a ascending)
template <class T, class X>
__device__ __forceinline__ uint64_t upper_bound_dev(const T *a, uint64_t lo, uint64_t hi, X x) {
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((X)a[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first index in [lo,hi) with a[idx] >= x
template <class T, class X>
__device__ __forceinline__ uint64_t lower_bound_dev(const T *a, uint64_t lo, uint64_t hi, X x) {
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((X)a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- validation (reference: dupcheck SimSpGEMM.cpp:43-53 + the ordering coo2csr guarantees) ----
__global__ void validate_ptr_kernel(const int64_t *ptr, uint64_t nseg, uint64_t nnz, uint32_t *flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nseg) return;
    int64_t v = ptr[i];
    bool bad = v < 0 || (uint64_t)v > nnz || (i == 0 && v != 0) || (i == nseg && (uint64_t)v != nnz);
    if (i < nseg && ptr[i + 1] < v) bad = true;
    if (bad) atomicOr(flags, kFlagPtr);
}
__global__ void validate_idx_kernel(const int64_t *ptr, const uint32_t *idx, uint64_t nseg,
                                    uint64_t nnz, uint64_t bound, uint32_t *flags) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    uint32_t v = idx[e];
    if (v >= bound) atomicOr(flags, kFlagRange);
    if (e + 1 < nnz) {
        uint32_t nx = idx[e + 1];
        if (nx <= v) {
            // only a violation when e and e+1 are in the same segment
            uint64_t seg = upper_bound_dev(ptr, 0, nseg + 1, (int64_t)e) - 1;
            if ((uint64_t)ptr[seg + 1] > e + 1) atomicOr(flags, nx == v ? kFlagDuplicate : kFlagUnsorted);
        }
    }
}

// ---- symbolic ----------------------------------------------------------------------------------
// For every non-zero e of A (CSC order, column k in [k0,k1)): key[e] = its row, payload = e,
// w[e] = nnz(B[k,:]) = the length of its chunk.
__global__ void sym_expand_kernel(const int64_t *a_colptr, const uint32_t *a_rowidx,
                                  const int64_t *b_rowptr, uint64_t k0, uint64_t k1, int64_t e0,
                                  uint64_t nnz, uint32_t *key, uint32_t *payload, uint32_t *w) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    int64_t e = e0 + (int64_t)t;
    uint64_t k = upper_bound_dev(a_colptr, k0, k1 + 1, e) - 1;
    key[t] = a_rowidx[e];
    payload[t] = (uint32_t)t;
    w[t] = (uint32_t)(b_rowptr[k + 1] - b_rowptr[k]);
}

struct LoadGatherW {  // w in row order
    const uint32_t *w;
    const uint32_t *perm;
    __device__ uint64_t operator()(uint64_t t) const { return w[perm[t]]; }
};

// chunk_off[perm[t]] = offs_sorted[t]
__global__ void sym_scatter_offsets_kernel(const uint32_t *perm, const uint64_t *offs_sorted,
                                           uint64_t nnz, uint64_t *chunk_off) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nnz) chunk_off[perm[t]] = offs_sorted[t];
}

// row_off[i] = staging offset of row i's first partial product, i in [0, M]
__global__ void sym_row_offsets_kernel(const uint32_t *rows_sorted, const uint64_t *offs_sorted,
                                       uint64_t nnz, uint64_t M, uint64_t *row_off) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    uint64_t t = lower_bound_dev(rows_sorted, 0, nnz, (uint64_t)i);
    row_off[i] = offs_sorted[t];  // offs_sorted has nnz+1 entries, [nnz] = P
}

// ---- per-panel column windows ------------------------------------------------------------------
// Panel = output rows [r0,r1).  For column k: the sub-column of A whose rows fall in the panel
// (contiguous, rows ascend inside a column) and its number of partial products.
__global__ void panel_columns_kernel(const int64_t *a_colptr, const uint32_t *a_rowidx,
                                     const int64_t *b_rowptr, uint64_t k0, uint64_t nk, uint32_t r0,
                                     uint64_t r1, int whole, int64_t *a_start, uint32_t *a_cnt,
                                     uint64_t *prod) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nk) return;
    uint64_t k = k0 + t;
    uint64_t lo = (uint64_t)a_colptr[k], hi = (uint64_t)a_colptr[k + 1];
    if (!whole) {
        uint64_t l2 = lower_bound_dev(a_rowidx, lo, hi, (uint64_t)r0);
        hi = lower_bound_dev(a_rowidx, l2, hi, r1);
        lo = l2;
    }
    a_start[t] = (int64_t)lo;
    a_cnt[t] = (uint32_t)(hi - lo);
    prod[t] = (hi - lo) * (uint64_t)(b_rowptr[k + 1] - b_rowptr[k]);
}
struct LoadU64 {
    const uint64_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};

// ---- multiply ----------------------------------------------------------------------------------
// Reference: cscMulcsr, SimSpGEMM.cpp:265-281.  The panel's products are numbered k-major,
// then by A entry j, then by B entry l; wave `wv` owns products [wv*kMulPerWave, ...).  For each
// column it touches, the B row is held in registers (one entry per lane) and every A entry's chunk
// is written with consecutive lanes on consecutive addresses.
template <class T>
__global__ __launch_bounds__(kMulThreads) void multiply_kernel(
    const T *__restrict__ a_vals, const uint32_t *__restrict__ b_colidx, const T *__restrict__ b_vals,
    const int64_t *__restrict__ b_rowptr, const uint64_t *__restrict__ chunk_off, int64_t e0,
    const int64_t *__restrict__ a_start, const uint32_t *__restrict__ a_cnt,
    const uint64_t *__restrict__ prod_off, uint64_t k0, uint64_t nk, uint64_t total, uint64_t base,
    uint32_t *__restrict__ pcol, T *__restrict__ pval) {
    const unsigned lane = lane_id();
    const uint64_t wv = (uint64_t)blockIdx.x * (kMulThreads / kWave) + (threadIdx.x >> 6);
    const uint64_t ws = wv * kMulPerWave;
    if (ws >= total) return;
    const uint64_t we = min(ws + (uint64_t)kMulPerWave, total);
    // first column with products beyond ws
    uint64_t kk = upper_bound_dev(prod_off, 0, nk + 1, ws) - 1;
    uint64_t cur = ws;
    while (cur < we) {
        // advance to the column that holds product `cur` (short linear probe, then bisect)
        for (int step = 0; prod_off[kk + 1] <= cur;) {
            if (++step > 4) { kk = upper_bound_dev(prod_off, kk + 1, nk + 1, cur) - 1; break; }
            kk++;
        }
        const uint64_t p0 = prod_off[kk], p1 = prod_off[kk + 1];
        const uint64_t k = k0 + kk;
        const uint64_t bs = (uint64_t)b_rowptr[k];
        const uint32_t nb = (uint32_t)((uint64_t)b_rowptr[k + 1] - bs);
        const uint64_t as = (uint64_t)a_start[kk];
        const uint64_t a = cur - p0;                    // first product of this column we own
        const uint64_t b = min(we, p1) - p0;            // one past the last
        uint64_t j0, j1, la, lb;                        // A entries [j0,j1], l window at the ends
        if (a == 0 && b == p1 - p0) {
            j0 = 0; j1 = a_cnt[kk] - 1; la = 0; lb = nb;
        } else {
            j0 = a / nb; la = a - j0 * nb;
            j1 = (b - 1) / nb; lb = b - j1 * nb;
        }
        if (nb > 32) {
            for (uint32_t l0 = 0; l0 < nb; l0 += kWave) {
                const uint32_t l = l0 + lane;
                const bool in = l < nb;
                uint32_t bc = 0; T bv = 0;
                if (in) { bc = b_colidx[bs + l]; bv = b_vals[bs + l]; }
                for (uint64_t j = j0; j <= j1; j++) {
                    const uint64_t e = as + j;
                    const T av = a_vals[e];
                    const uint64_t off = chunk_off[e - (uint64_t)e0] - base;
                    const bool ok = in && !(j == j0 && l < la) && !(j == j1 && l >= lb);
                    if (ok) { pcol[off + l] = bc; pval[off + l] = av * bv; }
                }
            }
        } else {
            // several A entries per wave instruction: lane -> (jj, l)
            const uint32_t g = kWave / nb;
            const uint32_t jj = (lane * ((65536u + nb - 1) / nb)) >> 16;  // lane / nb (exact, lane<64)
            const uint32_t l = lane - jj * nb;
            const bool in = jj < g;
            uint32_t bc = 0; T bv = 0;
            if (in) { bc = b_colidx[bs + l]; bv = b_vals[bs + l]; }
            for (uint64_t jb = j0; jb <= j1; jb += g) {
                const uint64_t j = jb + jj;
                bool ok = in && j <= j1 && !(j == j0 && l < la) && !(j == j1 && l >= lb);
                if (ok) {
                    const uint64_t e = as + j;
                    const uint64_t off = chunk_off[e - (uint64_t)e0] - base;
                    pcol[off + l] = bc;
                    pval[off + l] = a_vals[e] * bv;
                }
            }
        }
        cur = p0 + b;
    }
}

// ---- tile planning -----------------------------------------------------------------------------
// Row r of the panel starts a merge tile when its staging offset enters a new half-tile slot, when
// it or its predecessor is longer than half a tile, or every kTileMaxRows rows.
struct TileStartFlag {
    const uint64_t *row_off;
    uint64_t r0, r1, base;
    uint32_t max_rows;
    uint32_t kTileHalf;  // half the tile capacity
    __device__ uint32_t operator()(uint64_t t) const {
        const uint64_t r = r0 + t;
        if (t == 0) return 1;
        const uint64_t o_prev = row_off[r - 1], o = row_off[r], o_next = row_off[r + 1];
        const bool heavy = (o_next - o) > (uint64_t)kTileHalf, heavy_prev = (o - o_prev) > (uint64_t)kTileHalf;
        const bool slot = ((o - base) / kTileHalf) != ((o_prev - base) / kTileHalf);
        return (heavy || heavy_prev || slot || (t % max_rows) == 0) ? 1u : 0u;
    }
};
struct HeavyRowFlag {
    const uint64_t *row_off;
    uint64_t r0;
    uint32_t kTileCap;
    __device__ uint32_t operator()(uint64_t t) const {
        const uint64_t r = r0 + t;
        return (row_off[r + 1] - row_off[r]) > (uint64_t)kTileCap ? 1u : 0u;
    }
};
// list[scan[t]] = r0 + t for flagged t
template <class F>
__global__ void compact_flagged_kernel(F f, const uint32_t *scan, uint64_t n, uint64_t r0,
                                       uint32_t *list) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && f(t)) list[scan[t]] = (uint32_t)(r0 + t);
}

// ---- merge: LDS tile ---------------------------------------------------------------------------
// Reference: deduplicateCOO, SimSpGEMM.cpp:519-535 (sort by (row,col), sum equal keys, keep
// zeros).  One workgroup per tile.  The merged rows are written back over the tile's own span
// (it was fully staged into LDS first) and row_nnz / row_src record where each row now lives.
template <class T>
struct MergeSmem {
    static constexpr int kTileCap = TileCap<T>::value;
    uint32_t key[2][kTileCap];
    uint16_t pos[2][kTileCap];
    uint16_t rank[kTileCap + 1];
    T val[kTileCap];
    uint32_t cnt[kMergeThreads / kWave][kRadix];
    uint32_t rowo[kTileMaxRows + 1];
    uint32_t scratch[kMergeThreads / kWave + 1];
};

template <class T>
__global__ __launch_bounds__(kMergeThreads) void merge_tiles_kernel(
    const uint32_t *__restrict__ tile_rows, uint32_t ntiles, uint64_t r_end,
    const uint64_t *__restrict__ row_off, uint64_t base, int colbits, uint32_t *pcol, T *pval,
    uint32_t *__restrict__ row_nnz, uint64_t *__restrict__ row_src) {
    __shared__ MergeSmem<T> sm;
    constexpr int kTileCap = TileCap<T>::value;
    constexpr int NW = kMergeThreads / kWave;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const uint32_t t = blockIdx.x;
    const uint64_t ra = tile_rows[t];
    const uint64_t rb = (t + 1 < ntiles) ? (uint64_t)tile_rows[t + 1] : r_end;
    const uint32_t nr = (uint32_t)(rb - ra);
    const uint64_t s = row_off[ra] - base;
    const uint32_t n64 = (uint32_t)min(row_off[rb] - base - s, (uint64_t)kTileCap + 1);
    if (n64 > (uint32_t)kTileCap) return;  // a single long row: global-sort path
    const uint32_t n = n64;
    for (uint32_t r = tid; r <= nr; r += kMergeThreads) sm.rowo[r] = (uint32_t)(row_off[ra + r] - base - s);
    __syncthreads();
    if (n == 0) {
        for (uint32_t r = tid; r < nr; r += kMergeThreads) { row_nnz[ra + r] = 0; row_src[ra + r] = s; }
        return;
    }
    // stage: key = (local row << colbits) | col, payload = staging position
    for (uint32_t i = tid; i < n; i += kMergeThreads) {
        uint32_t lo = 0, hi = nr;  // last r with rowo[r] <= i
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (sm.rowo[mid] <= i) lo = mid; else hi = mid; }
        sm.key[0][i] = (colbits < 32 ? (lo << colbits) : 0u) | pcol[s + i];
        sm.pos[0][i] = (uint16_t)i;
        sm.val[i] = pval[s + i];
    }
    int rowbits = 0;
    while ((1u << rowbits) < nr) rowbits++;
    const int nbits = colbits + rowbits;
    // each wave ranks a contiguous quarter, so earlier waves = earlier positions (stable)
    const uint32_t per = (n + NW - 1) / NW;
    const uint32_t wbeg = min(w * per, n), wend = min(wbeg + per, n);
    int cur = 0;
    __syncthreads();
    for (int shift = 0; shift < nbits; shift += 8) {
        for (int d = lane; d < kRadix; d += kWave) sm.cnt[w][d] = 0;
        // (a) rank inside the wave's span
        for (uint32_t i0 = wbeg; i0 < wend; i0 += kWave) {
            const uint32_t i = i0 + lane;
            const bool valid = i < wend;
            const unsigned d = valid ? (sm.key[cur][i] >> shift) & 255u : 0u;
            const uint64_t peers = wave_match8(d, valid);
            const unsigned rk = __popcll(peers & lanemask_lt());
            if (valid) {
                const uint32_t c = sm.cnt[w][d];
                sm.rank[i] = (uint16_t)(c + rk);
                if (rk == 0) sm.cnt[w][d] = c + (uint32_t)__popcll(peers);
            }
        }
        __syncthreads();
        // (b) exclusive scan over (digit major, wave minor)
        {
            uint32_t c[NW], ssum = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { c[ww] = sm.cnt[ww][tid]; ssum += c[ww]; }
            uint32_t total;
            uint32_t ex = block_excl_scan<uint32_t, kMergeThreads>(ssum, sm.scratch, &total);
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { sm.cnt[ww][tid] = ex; ex += c[ww]; }
        }
        __syncthreads();
        // (c) scatter
        for (uint32_t i0 = wbeg; i0 < wend; i0 += kWave) {
            const uint32_t i = i0 + lane;
            if (i < wend) {
                const uint32_t k = sm.key[cur][i];
                const uint32_t dst = sm.cnt[w][(k >> shift) & 255u] + sm.rank[i];
                sm.key[cur ^ 1][dst] = k;
                sm.pos[cur ^ 1][dst] = sm.pos[cur][i];
            }
        }
        cur ^= 1;
        __syncthreads();
    }
    // head flags + exclusive scan (blocked: thread owns 16 consecutive sorted entries)
    constexpr int IPT = kTileCap / kMergeThreads;
    const uint32_t ib = tid * IPT;
    uint32_t heads = 0, hmask = 0;
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint32_t i = ib + q;
        if (i < n) {
            const bool h = (i == 0) || (sm.key[cur][i] != sm.key[cur][i - 1]);
            hmask |= (h ? 1u : 0u) << q;
            heads += h;
        }
    }
    uint32_t total;
    uint32_t ex = block_excl_scan<uint32_t, kMergeThreads>(heads, sm.scratch, &total);
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint32_t i = ib + q;
        if (i < n) {
            sm.rank[i] = (uint16_t)ex;  // output slot of the run that starts at/behind i
            ex += (hmask >> q) & 1u;
        }
    }
    if (tid == 0) sm.rank[n] = (uint16_t)total;
    __syncthreads();
    // each head sums its run in staging order and writes the merged entry in place
    const uint32_t colmask = colbits < 32 ? ((1u << colbits) - 1u) : 0xffffffffu;
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint32_t i = ib + q;
        if (i < n && ((hmask >> q) & 1u)) {
            const uint32_t k = sm.key[cur][i];
            T acc = sm.val[sm.pos[cur][i]];
            for (uint32_t u = i + 1; u < n && sm.key[cur][u] == k; u++) acc += sm.val[sm.pos[cur][u]];
            const uint32_t o = sm.rank[i];
            pcol[s + o] = k & colmask;
            pval[s + o] = acc;
        }
    }
    // rows keep their index span through the sort (row is the major key)
    for (uint32_t r = tid; r < nr; r += kMergeThreads) {
        const uint32_t x0 = sm.rowo[r], x1 = sm.rowo[r + 1];
        const uint32_t o0 = sm.rank[x0], o1 = sm.rank[x1];
        row_nnz[ra + r] = o1 - o0;
        row_src[ra + r] = s + o0;
    }
}

// ---- merge: global-sort path for rows longer than kTileCap ---------------------------------------
struct HeavyLen {
    const uint32_t *rows;
    const uint64_t *row_off;
    __device__ uint64_t operator()(uint64_t h) const { return row_off[rows[h] + 1] - row_off[rows[h]]; }
};
// key = (heavy rank << colbits) | col, payload = panel-relative staging position
__global__ void heavy_fill_kernel(const uint32_t *rows, const uint64_t *hoff, uint32_t nheavy,
                                  const uint64_t *row_off, uint64_t base, int colbits,
                                  const uint32_t *pcol, uint64_t nh, uint64_t *key, uint32_t *pos) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nh) return;
    uint64_t h = upper_bound_dev(hoff, 0, (uint64_t)nheavy + 1, x) - 1;
    uint64_t p = row_off[rows[h]] - base + (x - hoff[h]);
    key[x] = (h << colbits) | (uint64_t)pcol[p];
    pos[x] = (uint32_t)p;
}
struct HeavyHeadFlag {
    const uint64_t *key;
    __device__ uint32_t operator()(uint64_t x) const { return (x == 0 || key[x] != key[x - 1]) ? 1u : 0u; }
};
template <class T>
__global__ void heavy_gather_kernel(const uint32_t *pos, const T *pval, uint64_t nh, T *sorted_val) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < nh) sorted_val[x] = pval[pos[x]];
}
// every run head sums its run (ascending staging position = ascending k) and writes in place
template <class T>
__global__ void heavy_reduce_kernel(const uint64_t *key, const T *sorted_val, const uint64_t *headscan,
                                    uint64_t nh, const uint32_t *rows, const uint64_t *hoff,
                                    uint32_t nheavy, const uint64_t *row_off, uint64_t base,
                                    int colbits, uint32_t *pcol, T *pval) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nh) return;
    const uint64_t k = key[x];
    if (x != 0 && key[x - 1] == k) return;
    T acc = sorted_val[x];
    for (uint64_t u = x + 1; u < nh && key[u] == k; u++) acc += sorted_val[u];
    const uint64_t h = k >> colbits;
    const uint64_t o = row_off[rows[h]] - base + (headscan[x] - headscan[hoff[h]]);
    pcol[o] = (uint32_t)(k & ((1ull << colbits) - 1ull));
    pval[o] = acc;
}
__global__ void heavy_rows_kernel(const uint32_t *rows, const uint64_t *hoff, uint32_t nheavy,
                                  const uint64_t *headscan, const uint64_t *row_off, uint64_t base,
                                  uint32_t *row_nnz, uint64_t *row_src) {
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nheavy) return;
    row_nnz[rows[h]] = (uint32_t)(headscan[hoff[h + 1]] - headscan[hoff[h]]);
    row_src[rows[h]] = row_off[rows[h]] - base;
}

// ---- compaction --------------------------------------------------------------------------------
struct LoadRowNnz {
    const uint32_t *row_nnz;
    uint64_t r0;
    __device__ uint64_t operator()(uint64_t t) const { return row_nnz[r0 + t]; }
};
// out[c_rowptr_local[r] + i] = staging[row_src[r] + i]; thread owns 8 consecutive outputs
template <class T>
__global__ __launch_bounds__(256) void compact_rows_kernel(
    const uint64_t *__restrict__ c_local /* nr+1, exclusive scan of row_nnz */, uint64_t r0, uint64_t nr,
    const uint64_t *__restrict__ row_src, const uint32_t *__restrict__ pcol, const T *__restrict__ pval,
    uint64_t nnz, uint32_t *__restrict__ c_col, T *__restrict__ c_val) {
    constexpr int IPT = 8;
    const uint64_t o0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * IPT;
    if (o0 >= nnz) return;
    uint64_t r = upper_bound_dev(c_local, 0, nr + 1, o0) - 1;  // row containing output o0
    uint64_t rend = c_local[r + 1];
    uint64_t src = row_src[r0 + r] + (o0 - c_local[r]);
#pragma unroll
    for (int q = 0; q < IPT; q++) {
        const uint64_t o = o0 + q;
        if (o >= nnz) break;
        while (o >= rend) {  // next non-empty row
            r++;
            rend = c_local[r + 1];
            src = row_src[r0 + r];
        }
        c_col[o] = pcol[src];
        c_val[o] = pval[src];
        src++;
    }
}
__global__ void rowptr_finalize_kernel(const uint64_t *c_local, uint64_t nr, uint64_t offset,
                                       int64_t *c_rowptr /* + r0 */) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= nr) c_rowptr[t] = (int64_t)(c_local[t] + offset);
}

// ---- CSR parts -> staging (multi-GPU final merge) ----------------------------------
// Row r of part p becomes one chunk of row r; chunks ordered by p.
struct PartsRowLen {
    const int64_t *const *rowptrs;
    int nparts;
    __device__ uint64_t operator()(uint64_t r) const {
        uint64_t s = 0;
        for (int p = 0; p < nparts; p++) s += (uint64_t)(rowptrs[p][r + 1] - rowptrs[p][r]);
        return s;
    }
};
template <class T>
__global__ void parts_scatter_kernel(const int64_t *const *rowptrs, const uint32_t *const *colidxs,
                                     const T *const *valss, int nparts, uint64_t r0, uint64_t r1,
                                     const uint64_t *row_off, uint64_t base, uint32_t *pcol, T *pval) {
    // one wave per row
    const uint64_t r = r0 + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    if (r >= r1) return;
    uint64_t dst = row_off[r] - base;
    for (int p = 0; p < nparts; p++) {
        const int64_t b = rowptrs[p][r], e = rowptrs[p][r + 1];
        for (int64_t i = b + lane_id(); i < e; i += kWave) {
            pcol[dst + (i - b)] = colidxs[p][i];
            pval[dst + (i - b)] = valss[p][i];
        }
        dst += (uint64_t)(e - b);
    }
}

}  // namespace osp
