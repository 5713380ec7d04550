// osp_kernels.h -- HIP kernels of the outer-product SpGEMM for gfx950 (MI355X, wave64).
//
// Pipeline (host orchestration in osp_api.hip):
//   symbolic   every non-zero A[i,k] owns one CHUNK of nnz(B[k,:]) partial products; chunks of one
//              output row are laid out contiguously, ordered by k, so a row's partial products are
//              one contiguous span of the staging buffer and the row id is implicit.
//   plan       (osp_split.h) rows longer than one merge tile: their column ranges and, for every chunk
//              and range, where the chunk's run goes -- so that the multiply can write such a row
//              range by range ("direct" rows) instead of a split pass moving it afterwards.
//   multiply   cscMulcsr (SimSpGEMM.cpp:265-281): column k of A x row k of B, written chunk by
//              chunk (or run by run).  The product space is flattened and cut into equal slices,
//              one per wave.
//   merge      deduplicateCOO (SimSpGEMM.cpp:519-535): per tile of consecutive rows -- or ranges of one
//              long row --, a stable LSD radix sort on (row, col) in LDS, equal keys summed in staging
//              order (= ascending k, the order the oracle's stable sort yields), every tile placed in
//              the final CSR by a decoupled look-back: merged rows are written once.
#pragma once
#include <type_traits>
#include "osp_prims.h"

namespace osp {

// ---- tunables --------------------------------------------------------------------------------
// partial products one LDS merge tile holds.  Six per thread of a 256-thread workgroup: the kernel then needs 96
// registers, so FIVE workgroups run per CU (31 KB of LDS each).  (Measured, tools/bench_merge, 2.7e8 partial
// products: two workgroups of 512 threads on 3072-entry tiles 4.0 ms, four of 256 on 1536 3.5 ms, five 3.3 ms --
// independent barrier domains per CU matter more than tile size, although every tile costs a look-back.  Round 2, same
// tool, level-1 tiles at full fill: 1536 x 256 threads x 5 per CU 2.40 ms; 1792 x 256 x 4 2.65; 2048 x 256 x 4 2.75;
// 1920 x 320 x 4 3.30; 2304 x 384 x 3 3.96; 3072 x 512 x 2 3.22.)
template <class T> struct TileCap;
#ifndef OSP_TILE_CAP_F32
#define OSP_TILE_CAP_F32 1792
#endif
// (f32 values alias 4 bytes per entry, not 8: the same 31 KB of LDS hold a seventh more entries, and a tile costs a ticket
// and a look-back whatever it holds)
template <> struct TileCap<float> { static constexpr int value = OSP_TILE_CAP_F32; };
#ifndef OSP_TILE_CAP_F64
#define OSP_TILE_CAP_F64 1536
#endif
template <> struct TileCap<double> { static constexpr int value = OSP_TILE_CAP_F64; };
constexpr int kMergeThreads = 256;
constexpr int kTileMaxRows = kMergeThreads - 1;  // rows per tile: 8 row bits in the sort key, one row offset per thread
constexpr int kMulThreads = 256;
#ifndef OSP_MUL_PER_WAVE
#define OSP_MUL_PER_WAVE 2048
#endif
constexpr int kMulPerWave = OSP_MUL_PER_WAVE;  // partial products per wave slice
constexpr int kMulPerBlock = kMulPerWave * (kMulThreads / kWave);
#ifndef OSP_MUL_BATCH_MAX
#define OSP_MUL_BATCH_MAX 4096
#endif
constexpr int kMulBatchMax = OSP_MUL_BATCH_MAX;  // up to 64 consecutive columns with at most this many products are one batch
constexpr int kMulBatchAvg = 64;  // ... and at most this many per column on average
#ifndef OSP_MUL_TILE_MIN
#define OSP_MUL_TILE_MIN 1024
#endif
// chunks whose destinations multiply_kernel forms together (chunk_dests), per path.  Measured on R-MAT-22: 1, 2, 4 or 8 at a
// time make no difference to the direct rows (the kernel is bound by partially written lines, not by the gathers), while 4 / 2
// cost 20 registers and two waves per SIMD, which the products WITHOUT long rows pay for (uniform: 8.0 -> 8.9 ms).
#ifndef OSP_MUL_QU_MID
#define OSP_MUL_QU_MID 2
#endif
#ifndef OSP_MUL_QU_HUB
#define OSP_MUL_QU_HUB 1
#endif
constexpr int kMulTileMin = OSP_MUL_TILE_MIN;  // B rows from this length on: products numbered in panels of the row (multiply_kernel)
constexpr int kMulTileW = 128;                 // ... of this many entries (256: 77 registers, slower everywhere; 64: slower at Graph500 skew)

// One staged partial product: 4-byte column + value, packed (12 B for f64, 8 B for f32).  Array of
// records rather than two arrays: a chunk is then ONE contiguous byte range, which halves the number
// of partially written cache lines per chunk (tools/bench_scatter: 16-entry chunks at unaligned
// offsets go from 1.0 to 1.8 TB/s).
template <class T>
struct __attribute__((packed, aligned(4))) Part {
    uint32_t col;
    T val;
};

// A record as it sits in registers right after the load: raw 32-bit words.  Unpacking a 12-byte record into
// (col, aligned 64-bit val) moves registers, and the compiler waits for the load at that move -- a batch of
// loads written as `Part<T> p = stage[i]` degenerates into a chain of HBM round trips (seen in the ISA: one
// s_waitcnt vmcnt(0) per load).  So batches are loaded raw and unpacked where they are used.
template <class T>
struct __attribute__((packed, aligned(4))) PartWords {
    uint32_t w[sizeof(Part<T>) / 4];
    __device__ __forceinline__ uint32_t col() const { return w[0]; }
    __device__ __forceinline__ T val() const {
        if constexpr (sizeof(T) == 8) return (T)__hiloint2double((int)w[2], (int)w[1]);
        else return (T)__uint_as_float(w[1]);
    }
};
template <class T>
__device__ __forceinline__ PartWords<T> load_part_words(const Part<T> *p) { return *reinterpret_cast<const PartWords<T> *>(p); }
template <class T>
__device__ __forceinline__ void store_part_words(Part<T> *p, const PartWords<T> &r) { *reinterpret_cast<PartWords<T> *>(p) = r; }

// Streaming store of one record (written once, read much later by another kernel): non-temporal, so that the staging
// buffer does not push the operands out of L2.  OSP_NT_STAGE=0 compiles the plain store (A/B measurements: multiply
// 58 -> 48 ms on the default workload together with the non-temporal stores of the merge output).  The same hint on
// the split kernels' scattered stores made them much slower (they need L2 to combine partial lines: 333 -> 394 ms);
// on loads at their last use it made no difference.
#ifndef OSP_NT_STAGE
#define OSP_NT_STAGE 1
#endif
template <class T>
__device__ __forceinline__ void stream_store_part(Part<T> *p, uint32_t col, T val) {
#if OSP_NT_STAGE
    if constexpr (sizeof(T) == 8) {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
        const uint64_t b = (uint64_t)__double_as_longlong((double)val);
        u32x3 v;
        v.x = col; v.y = (uint32_t)b; v.z = (uint32_t)(b >> 32);
        __builtin_nontemporal_store(v, reinterpret_cast<u32x3 *>(p));
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
        u32x2 v;
        v.x = col; v.y = __float_as_uint((float)val);
        __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p));
    }
#else
    *p = Part<T>{col, val};
#endif
}

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
// upper_bound_dev by a whole wave: 64 probes per step (a 2^22 range takes 4 round trips instead of 22); every lane of the
// wave must call it with the same arguments, every lane gets the result
template <class T, class X>
__device__ __forceinline__ uint64_t wave_upper_bound(const T *a, uint64_t lo, uint64_t hi, X x) {
    const uint64_t lane = lane_id();
    while (hi - lo > 64) {
        const uint64_t step = (hi - lo) / 64, p = lo + lane * step;   // probes lo, lo + step, ... (all below hi)
        const int c = __popcll(__ballot((X)a[p] <= x));               // monotone: the first c probes hold values <= x
        if (c == 0) return lo;
        if (c < 64) hi = lo + (uint64_t)c * step;
        lo = lo + (uint64_t)(c - 1) * step + 1;
    }
    const bool le = lane < hi - lo && (X)a[lo + lane] <= x;
    return lo + (uint64_t)__popcll(__ballot(le));
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

// columns of packed records (osp_merge_record_parts with cfg.validate): every one below its dimension
template <class T>
__global__ void validate_record_cols_kernel(const Part<T> *rec, uint64_t n, uint64_t bound, uint32_t *flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (uint64_t)rec[i].col >= bound) atomicOr(flags, kFlagRange);
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

// ---- the row-wise variant -----------------------------------------------------------
// cfg.algorithm = OSP_ALGO_ROWWISE: rows whose partial products fit one merge tile are never staged.  The tile kernel
// computes them itself, row by row, from A's row (the chunk table below: one entry per non-zero A[i,k]) and the B rows
// it touches -- Gustavson's formulation, with the tile's sort + run sums in place of a hash accumulator, so that
// values stay bit-identical -- and the multiply phase skips those rows (kChunkSkip).  Longer rows keep the
// outer-product path: staged, split, merged.
constexpr uint64_t kChunkSkip = ~0ull;  // chunk_off of a chunk the multiply phase must not write
template <class T>
struct ChunkTable {
    const uint64_t *off;       // staging offset of chunk t (nnz + 1 entries), chunks in (row, k) order
    const uint32_t *bs;        // first entry of the chunk's B row in b_colidx / b_vals
    const uint32_t *perm;      // the chunk's A entry: position in a_vals (already offset to the k shard)
    const uint32_t *rowfirst;  // first chunk of every row (M + 1 entries)
    const T *a_vals;
    const uint32_t *b_colidx;
    const T *b_vals;
    uint32_t cap;              // rows of up to `cap` partial products are computed row-wise
    uint32_t enabled;
};
// chunk_off[perm[t]] = offs_sorted[t], or kChunkSkip for the chunks of rows computed row-wise
__global__ void sym_scatter_offsets_kernel(const uint32_t *perm, const uint64_t *offs_sorted, const uint32_t *rows_sorted,
                                           const uint64_t *row_off, uint64_t rowwise_cap, uint64_t nnz, uint64_t *chunk_off) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const uint32_t r = rows_sorted[t];
    chunk_off[perm[t]] = (row_off[r + 1] - row_off[r]) <= rowwise_cap ? kChunkSkip : offs_sorted[t];
}

// row_off[i] = staging offset of row i's first partial product, arow[i] = index of its first
// non-empty chunk, i in [0, M]
__global__ void sym_row_offsets_kernel(const uint32_t *rows_sorted, const uint64_t *offs_sorted, uint64_t nnz, uint64_t M,
                                       uint64_t *row_off, uint32_t *rowfirst) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    uint64_t t = lower_bound_dev(rows_sorted, 0, nnz, (uint64_t)i);
    row_off[i] = offs_sorted[t];  // offs_sorted has nnz+1 entries, [nnz] = P
    rowfirst[i] = (uint32_t)t;    // row i's chunks are entries [rowfirst[i], rowfirst[i+1]) of the (row, k) order
}
struct LenGatherW {  // chunk length of the t-th A entry in (row, k) order
    const uint32_t *w;
    const uint32_t *perm;
    __device__ uint32_t operator()(uint64_t t) const { return w[perm[t]]; }
};

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
// The row panels of a product that does not fit the staging capacity (ONE wave): consecutive rows whose partial products
// fit `cap`, never across one of `cuts` (ascending absolute row ids; the multi-GPU merge ends its panels where the pieces it
// receives end).  out[0] = number of panels, out[1] = ~0 or the first row that alone exceeds the capacity (or the row at which
// maxp panels were used up), out[2 + p] = first row of panel p (and r_lo + M behind the last), out[2 + maxp + 1 + p] = its staging offset.
__global__ void panel_bounds_kernel(const uint64_t *row_off, uint64_t r_lo, uint64_t M, uint64_t cap, const uint64_t *cuts, uint32_t ncuts,
                                    uint32_t maxp, uint64_t *out) {
    const uint64_t *off = row_off + r_lo;
    uint64_t r = 0, bad = ~0ull;
    uint32_t p = 0, ci = 0;
    if (lane_id() == 0) { out[2] = r_lo; out[2 + maxp + 1] = off[0]; }
    while (r < M) {
        if (p == maxp) { bad = r_lo + r; break; }
        // largest r1 with off[r1] - off[r] <= cap
        uint64_t r1 = wave_upper_bound(off, r, M + 1, off[r] + cap) - 1;
        if (r1 <= r) { bad = r_lo + r; break; }
        r1 = min(r1, M);
        while (ci < ncuts && cuts[ci] <= r_lo + r) ci++;
        if (ci < ncuts && cuts[ci] < r_lo + r1) r1 = cuts[ci] - r_lo;
        p++;
        if (lane_id() == 0) { out[2 + p] = r_lo + r1; out[2 + maxp + 1 + p] = off[r1]; }
        r = r1;
    }
    if (lane_id() == 0) { out[0] = p; out[1] = bad; }
}
struct LoadU64 {
    const uint64_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
// A panel with gathered rows multiplies a compacted A (multiply_kernel, IND): entry e0 + t is KEPT when its row is in the
// panel and its chunk is written (not kChunkSkip); elist = the kept entries in CSC order, and per column the window of the list.
// (desc_only: only the chunks that carry a descriptor -- rows written through cells; the plainly staged rows of such a panel
// are expanded row by row, expand_rows_kernel)
struct PanelKeepFlag {
    const uint32_t *a_rowidx; const uint64_t *chunk_off; int64_t e0; uint32_t r0; uint64_t r1; uint32_t desc_only;
    __device__ uint32_t operator()(uint64_t t) const {
        const uint32_t r = a_rowidx[(uint64_t)e0 + t];
        if (!(r >= r0 && (uint64_t)r < r1)) return 0u;
        const uint64_t off = chunk_off[t];
        return (off != ~0ull && (!desc_only || (off >> 63))) ? 1u : 0u;
    }
};
__global__ void panel_keep_list_kernel(PanelKeepFlag f, const uint32_t *kscan, uint64_t nnz, uint32_t *elist) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nnz && f(t)) elist[kscan[t]] = (uint32_t)((uint64_t)f.e0 + t);
}
// ... the columns that keep entries, compacted: column number cscan[t] of the list is k0 + t (cscan: exclusive scan of the flags)
struct PanelKeepColFlag {
    const int64_t *a_colptr; int64_t e0; uint64_t k0; const uint32_t *kscan;
    __device__ uint32_t operator()(uint64_t t) const { return kscan[a_colptr[k0 + t + 1] - e0] != kscan[a_colptr[k0 + t] - e0] ? 1u : 0u; }
};
__global__ void panel_keep_columns_kernel(const int64_t *a_colptr, const int64_t *b_rowptr, uint64_t k0, uint64_t nk, int64_t e0,
                                          const uint32_t *kscan, const uint32_t *cscan, int64_t *a_start, uint32_t *a_cnt, uint64_t *prod,
                                          uint32_t *klist) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nk) return;
    const uint64_t k = k0 + t;
    const uint32_t lo = kscan[a_colptr[k] - e0], hi = kscan[a_colptr[k + 1] - e0];
    const uint32_t ncol = cscan[nk];   // columns in the list; the entries behind them: no products, column k0
    if (t >= ncol) { a_start[t] = 0; a_cnt[t] = 0; prod[t] = 0; klist[t] = (uint32_t)k0; }
    if (hi == lo) return;
    const uint32_t j = cscan[t];
    a_start[j] = (int64_t)lo;
    a_cnt[j] = hi - lo;
    prod[j] = (uint64_t)(hi - lo) * (uint64_t)(b_rowptr[k + 1] - b_rowptr[k]);
    klist[j] = (uint32_t)k;
}

// value of lane `src` (wave-uniform index) for every lane
__device__ __forceinline__ uint32_t wave_bcast(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }
__device__ __forceinline__ uint64_t wave_bcast(uint64_t v, uint32_t src) {
    return ((uint64_t)wave_bcast((uint32_t)(v >> 32), src) << 32) | wave_bcast((uint32_t)v, src);
}
__device__ __forceinline__ float wave_bcast(float v, uint32_t src) { return __uint_as_float(wave_bcast(__float_as_uint(v), src)); }
__device__ __forceinline__ double wave_bcast(double v, uint32_t src) {
    return __longlong_as_double((long long)wave_bcast((uint64_t)__double_as_longlong(v), src));
}

// ---- long rows written straight into their column ranges ("direct" rows) ----------------------------------------------
// A long output row (more partial products than one merge tile) must reach the tile kernel cut into column ranges.  The
// split kernels (osp_split.h) do that with an extra read-read-write pass over the row's records.  For a DIRECT row the
// multiply phase itself puts every product where the split would have moved it: the planner (direct_plan_kernel,
// osp_split.h) cuts the row's columns into ranges of at most one tile from an exact histogram, and stores for every chunk
// (non-zero A[i,k]) and range the position of the chunk's run inside that range -- ranges in column order, inside a range
// the runs in chunk order (ascending k), inside a run ascending columns: exactly the order the stable split produces, so
// the sums keep their bits.  chunk_off[e] of such a chunk is a descriptor instead of a staging offset:
//   bit 63 | shift (5 bits) | first cell of the chunk, in words from the row's block (26 bits) | the row's block (32 bits)
// and the row's block in `cells` is   [ range of every fine column bin: nfine bytes ][ chunk 0: one word per range ] ...
// A product (chunk, l-th entry of B's row, column c) goes to record  cell[range(c >> shift)] + l  of the second buffer
// (32-bit arithmetic; the cell holds the run's start minus the index of the run's first entry).
constexpr uint64_t kDirectBit = 1ull << 63;
__device__ __forceinline__ uint64_t direct_desc(uint32_t rowbase, uint32_t celloff, uint32_t sh) {
    return kDirectBit | ((uint64_t)(sh & 31u) << 58) | ((uint64_t)(celloff & 0x3ffffffu) << 32) | (uint64_t)rowbase;
}
// HUB rows (round 4): a row too long for the planner above (more than 128 K partial products: the hub rows of a skewed
// matrix) used to be moved range by range AFTER the multiply by the stretch split (osp_split.h): two more reads and one more
// write of every record.  Such a row's ranges are uniform blocks of 2^sh columns, so which of its entries a chunk puts into
// which block depends on B's row alone: a table made once per product numbers the RUNS of every B row (maximal sequences of
// entries inside one block; sx[e] = runs that start before entry e, over all of B) and the hub planner (hub_plan_kernel) stores
// one cell per (chunk, run): the run's place in the second buffer minus the index of its first entry.  The chunk's descriptor
// carries the marker 31 where a direct row's carries its shift, and the first cell's index in the low 58 bits; a product
// (chunk, l-th entry of B's row at position p) goes to record  hubcells[first cell + (sx[p + 1] - 1 - sx[row start])] + l.
constexpr uint64_t kHubCellMask = (1ull << 58) - 1ull;
__device__ __forceinline__ uint64_t hub_desc(uint64_t first_cell) { return kDirectBit | (31ull << 58) | (first_cell & kHubCellMask); }
__device__ __forceinline__ bool is_hub_desc(uint64_t desc) { return ((uint32_t)(desc >> 58) & 31u) == 31u; }   // (of a direct descriptor that is not kChunkSkip)
struct HubArgs {
    const uint32_t *cells = nullptr;   // one per (chunk of a hub row, run of its B row)
    const uint32_t *sx = nullptr;      // runs of B that start before entry e (nnz(B) + 1 entries)
};
// A record of a direct row: OSP_NT_DIRECT=1 streams it past L2 like the chunk-major records; 0 leaves it to L2, where the
// runs of consecutive chunks of a hub row -- adjacent in the range, written close in time -- can meet in one line.
#ifndef OSP_NT_DIRECT
#define OSP_NT_DIRECT 1
#endif
template <class T>
__device__ __forceinline__ void store_direct_part(Part<T> *p, uint32_t col, T val) {
#if OSP_NT_DIRECT
    stream_store_part(p, col, val);
#else
    *p = Part<T>{col, val};
#endif
}
// A record of a hub row: plain (OSP_NT_HUB=0) -- the runs that consecutive chunks of the row write into one block are adjacent
// and written close in time, so partial lines meet in L2 (Graph500 scale 20: multiply 24.2 -> 23.4 ms per launch with plain
// stores for hub AND direct rows, while the direct rows alone lose by them: R-MAT-22 mild 21.8 -> 25.1).
#ifndef OSP_NT_HUB
#define OSP_NT_HUB 0
#endif
template <class T>
__device__ __forceinline__ void store_hub_part(Part<T> *p, uint32_t col, T val) {
#if OSP_NT_HUB
    stream_store_part(p, col, val);
#else
    *p = Part<T>{col, val};
#endif
}
// one record to where chunk_dests put it (off: the chunk's offset or descriptor, wave-uniform)
template <class T, int MODE>
__device__ __forceinline__ void store_dest(Part<T> *p, uint64_t off, bool dir, uint32_t col, T val) {
    if (!dir) stream_store_part(p, col, val);
    else if (MODE == 2 && is_hub_desc(off)) store_hub_part(p, col, val);
    else store_direct_part(p, col, val);
}
// (bs: where the chunk's B row starts in B's arrays; the entry is bs + l)
template <class T, int MODE>
__device__ __forceinline__ void store_direct(const uint32_t *__restrict__ cells, const HubArgs &hub, Part<T> *__restrict__ qstage, uint64_t desc,
                                             uint64_t bs, uint32_t l, uint32_t bc, T v) {
    if constexpr (MODE == 2) {
        if (is_hub_desc(desc)) {
            const uint32_t ri = hub.sx[bs + l + 1] - 1u - hub.sx[bs];
            const uint32_t cell = hub.cells[(desc & kHubCellMask) + ri];
            store_hub_part(&qstage[(uint32_t)(cell + l)], bc, v);
            return;
        }
    }
    const uint32_t *rb = cells + (uint32_t)desc;
    const uint32_t sh = (uint32_t)(desc >> 58) & 31u, co = (uint32_t)(desc >> 32) & 0x3ffffffu;
    const uint32_t rg = reinterpret_cast<const uint8_t *>(rb)[bc >> sh];
    const uint32_t delta = rb[co + rg];
    store_direct_part(&qstage[(uint32_t)(delta + l)], bc, v);
}

// Where ONE entry of B's row (index ld in the row, column bc) goes in QU chunks at once.  off[i]: the chunk's staging
// offset (wave-uniform; the entry's position in the chunk is lp behind it) or its descriptor.  A direct chunk costs two
// dependent loads per entry (range of the column, cell of the range); written one chunk after the other, every store
// waits for both, and that latency -- not bandwidth -- bounds the kernel (R-MAT-22: 10.1 -> 17.3 ms per launch).  Here the
// loads of all QU chunks go out together, branch-free (a chunk that is not direct reads a word of `safe` and ignores it).
// MODE 2: the launch has hub rows too; ri = the entry's run number inside its B row (the same for every chunk of the column)
template <class T, int QU, int MODE>
__device__ __forceinline__ void chunk_dests(const uint32_t *__restrict__ cells, const HubArgs &hub, uint32_t ri, const void *safe,
                                            Part<T> *__restrict__ stage,
                                            Part<T> *__restrict__ qstage, const uint64_t (&off)[QU], uint32_t lp, uint32_t ld,
                                            uint32_t bc, Part<T> *(&dst)[QU], bool (&dir)[QU]) {
    constexpr bool DIRECT = MODE != 0;
    if constexpr (!DIRECT) {   // a launch without direct rows: every chunk has a plain staging offset
#pragma unroll
        for (int i = 0; i < QU; i++) { dir[i] = false; dst[i] = &stage[off[i] + lp]; }
        return;
    }
    const uint32_t *rb[QU];
    uint32_t rg[QU], delta[QU];
    bool hubc[QU];
#pragma unroll
    for (int i = 0; i < QU; i++) {
        dir[i] = (off[i] & kDirectBit) && off[i] != kChunkSkip;   // (wave-uniform)
        hubc[i] = MODE == 2 && dir[i] && is_hub_desc(off[i]);
        rb[i] = (dir[i] && !hubc[i]) ? cells + (uint32_t)off[i] : reinterpret_cast<const uint32_t *>(safe);
        const uint32_t sh = (uint32_t)(off[i] >> 58) & 31u;
        rg[i] = reinterpret_cast<const uint8_t *>(rb[i])[(dir[i] && !hubc[i]) ? bc >> sh : 0u];
#ifdef OSP_EXP_EXTRA_GATHER   // experiment only: one more gather of the same kind per store (is the kernel bound by its gathers?)
        {
            const uint32_t extra = reinterpret_cast<const uint8_t *>(rb[i])[dir[i] ? (bc >> sh) ^ 1u : 0u];
            asm volatile("" ::"v"(extra));
        }
#endif
    }
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const uint32_t co = (uint32_t)(off[i] >> 32) & 0x3ffffffu;
        const uint32_t *p = rb[i] + ((dir[i] && !hubc[i]) ? co + rg[i] : 0u);
        if constexpr (MODE == 2) { if (hubc[i]) p = hub.cells + ((off[i] & kHubCellMask) + ri); }   // one cell per (chunk, run of B's row)
        delta[i] = *p;
    }
#pragma unroll
    for (int i = 0; i < QU; i++) dst[i] = dir[i] ? &qstage[(uint32_t)(delta[i] + ld)] : &stage[off[i] + lp];
}

// ---- multiply ----------------------------------------------------------------------------------
// Reference: cscMulcsr, SimSpGEMM.cpp:265-281.  The panel's products are numbered k-major,
// then by A entry j, then by B entry l; wave `wv` owns products [wv*kMulPerWave, ...).  For each
// column it touches, the B row is held in registers (one entry per lane) and every A entry's chunk
// is written with consecutive lanes on consecutive addresses.
// DIRECT: the launch has direct rows (descriptors among the chunk offsets); without them the kernel is instantiated without
// that code (64 registers instead of 68: one more wave per SIMD, which the products of short rows notice)
// MODE: 0 = no direct rows in the launch, 1 = direct rows, 2 = direct rows and hub rows (descriptors with the hub marker)
// IND: a panel with gathered rows (their chunks are never written: most of the headline's products).  Walking the skipped
// chunks costs the multiply what writing them would in instructions (11 ms per launch for a seventh of the records), so such a
// panel multiplies a COMPACTED A: `elist` holds, column after column, the entries of A whose chunks are written (panel_keep_*
// below); a_start / a_cnt / prod_off count in that list, and `total` is read from the scan (the launch covers the panel's
// full count: the workgroups beyond the written products leave at once).
template <class T, int MODE = 1, bool IND = false>
__global__ __launch_bounds__(kMulThreads) void multiply_kernel(
    const T *__restrict__ a_vals, const uint32_t *__restrict__ b_colidx, const T *__restrict__ b_vals,
    const int64_t *__restrict__ b_rowptr, const uint64_t *__restrict__ chunk_off, int64_t e0,
    const int64_t *__restrict__ a_start, const uint32_t *__restrict__ a_cnt,
    const uint64_t *__restrict__ prod_off, uint64_t k0, uint64_t nk, uint64_t total, uint64_t base,
    Part<T> *__restrict__ stage, const uint32_t *__restrict__ cells = nullptr, Part<T> *__restrict__ qstage = nullptr,
    const HubArgs hub = HubArgs{}, const uint32_t *__restrict__ elist = nullptr, const uint32_t *__restrict__ klist = nullptr) {
    constexpr bool DIRECT = MODE != 0;
    const unsigned lane = lane_id();
    // (IND: the host does not know how many products are written -- a count on the device -- and a grid for the panel's full
    // count would be workgroups that leave at once, half a million of them: 4 ms of dispatch for nothing.  A fixed grid
    // strides over the slices instead.)
    if constexpr (IND) total = prod_off[nk];
    for (uint64_t wv = (uint64_t)blockIdx.x * (kMulThreads / kWave) + (threadIdx.x >> 6);; wv += (uint64_t)gridDim.x * (kMulThreads / kWave)) {
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
        {
            // Many small columns in a row (power-law graphs: most columns hold a handful of products): walking them one
            // by one is a chain of dependent loads per column.  Up to 64 consecutive columns that together hold at most
            // kMulBatchMax products are taken as one batch: lane q loads column kk+q's descriptors in ONE round and the
            // lanes walk the batch's products, each finding its column by bisecting the start offsets with ds_bpermute.
            // lane q: products of columns [kk, kk+q] -- the batch is the longest prefix that stays within the limit, so a
            // hub column ends a batch instead of spoiling it
            const uint64_t pe = prod_off[min(kk + lane + 1, nk)] - p0;
            // (a column whose B row is numbered in panels -- below -- never joins a batch: the waves that share a column
            // must agree on the numbering of its products, and the batch walk numbers them chunk by chunk)
            // (IND: the columns are the panel's columns WITH written products, klist names them -- between two of them lie
            // thousands that have none, and finding the next one through their zero counts cost a bisection of 22 dependent loads)
            uint64_t kcol = k0 + min(kk + lane, nk - 1);
            if constexpr (IND) kcol = klist[min(kk + lane, nk - 1)];
            const uint64_t bs_l = (uint64_t)b_rowptr[kcol];
            const uint32_t nb_l = (uint32_t)((uint64_t)b_rowptr[kcol + 1] - bs_l);
            const uint64_t fits = __ballot(kk + lane < nk && pe <= (uint64_t)kMulBatchMax && nb_l < (uint32_t)kMulTileMin);
            const uint32_t ncol = fits == ~0ull ? (uint32_t)kWave : (uint32_t)__builtin_ctzll(~fits);  // pe ascends: a prefix of lanes
            // (and only where the columns are small on average: with B's row in registers the per-column path below is the
            // faster one from about 64 products per column on -- uniform R-MAT measured 9.0 against 9.8 ms)
            if (ncol >= 2 && wave_bcast(pe, ncol - 1) <= (uint64_t)kMulBatchAvg * ncol) {
                const uint64_t kend = kk + ncol;
                const uint64_t pend = p0 + wave_bcast(pe, ncol - 1);
                const uint64_t kq = kk + lane;
                const bool cv = kq < kend;
                const uint32_t prev = (uint32_t)__shfl_up((int)(uint32_t)min(pe, (uint64_t)0x7fffffffu), 1);
                const uint32_t rel0 = cv ? (lane ? prev : 0u) : (uint32_t)(pend - p0);  // lanes past the end: behind all
                const uint64_t bsq = cv ? bs_l : 0ull;
                const uint32_t nbq = cv ? nb_l : 1u;
                const uint64_t asq = cv ? (uint64_t)a_start[kq] : 0ull;
                const uint32_t lo_p = (uint32_t)(cur - p0), hi_p = (uint32_t)(min(we, pend) - p0);
                for (uint32_t x0 = lo_p; x0 < hi_p; x0 += kWave) {
                    const uint32_t x = x0 + lane;
                    uint32_t lo = 0, hi = kWave;  // last lane whose column starts at or before x (empty columns never win)
#pragma unroll
                    for (int step = 0; step < 6; step++) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((uint32_t)__shfl((int)rel0, (int)mid) <= x) lo = mid; else hi = mid;
                    }
                    const uint32_t r = x - (uint32_t)__shfl((int)rel0, (int)lo);
                    const uint32_t nbx = (uint32_t)__shfl((int)nbq, (int)lo);
                    const uint64_t bsx = (uint64_t)__shfl((long long)bsq, (int)lo);
                    const uint64_t asx = (uint64_t)__shfl((long long)asq, (int)lo);
                    if (x < hi_p) {
                        const uint32_t j = r / nbx, l = r - j * nbx;
                        uint64_t e = asx + j;
                        if constexpr (IND) e = elist[e];
                        const uint64_t raw = chunk_off[e - (uint64_t)e0];
                        const uint32_t bc = b_colidx[bsx + l];
                        const T pv = a_vals[e] * b_vals[bsx + l];
                        if (raw != kChunkSkip) {
                            if (DIRECT && (raw & kDirectBit)) store_direct<T, MODE>(cells, hub, qstage, raw, bsx, l, bc, pv);
                            else stream_store_part(&stage[raw - base + l], bc, pv);
                        }
                    }
                }
                cur = p0 + hi_p;
                kk = kend - 1;
                continue;
            }
        }
        uint64_t k = k0 + kk;
        if constexpr (IND) k = klist[kk];
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
        if (nb >= (uint32_t)kMulTileMin) {
            // A long B row (a hub column's row at Graph500 skew holds 10^5 entries): numbered j-major, a wave's slice is a
            // piece of ONE chunk, so every product costs a 12-byte read of B beside its 12-byte write, each window of the row
            // fetched again by every wave that passes it.  The products of such a column are numbered in PANELS of
            // kMulTileW entries of the B row instead -- panel, then A entry, then entry inside the panel -- so that a slice
            // covers one window of the row for several A entries: the window is loaded once into registers (four entries
            // per lane) and written to every chunk of the slice.  Which wave writes a product changes, not where it goes.
            constexpr uint32_t W = (uint32_t)kMulTileW, PER = W / kWave;
            const uint64_t full = (uint64_t)a_cnt[kk] * W;   // products of a whole panel
            const uint32_t npanel = (nb + W - 1) / W;
            for (uint64_t x = a; x < b;) {
                const uint32_t pnl = (uint32_t)min(x / full, (uint64_t)npanel - 1);
                const uint32_t wp = pnl == npanel - 1 ? nb - pnl * W : W;
                const uint64_t pbeg = (uint64_t)pnl * full;
                const uint64_t xe = min(b, pbeg + (uint64_t)a_cnt[kk] * wp) - pbeg, xin = x - pbeg;
                const uint64_t tj0 = xin / wp, tj1 = (xe - 1) / wp;
                const uint32_t tla = (uint32_t)(xin - tj0 * wp), tlb = (uint32_t)(xe - tj1 * wp);
                uint32_t bc[PER], ri[PER];
                T bv[PER];
#pragma unroll
                for (uint32_t u = 0; u < PER; u++) {
                    const uint32_t lrel = u * kWave + lane;
                    const uint64_t src = bs + (uint64_t)pnl * W + min(lrel, wp - 1);   // clamped: the loads go out together
                    bc[u] = b_colidx[src];
                    bv[u] = b_vals[src];
                    ri[u] = 0;
                    if constexpr (MODE == 2) ri[u] = hub.sx[src + 1] - 1u - hub.sx[bs];   // the entry's run inside B's row (hub rows)
                }
                for (uint64_t jb = tj0; jb <= tj1; jb += kWave) {
                    const uint64_t jm = jb + lane;
                    T av_l = 0;
                    uint64_t off_l = kChunkSkip;
                    if (jm <= tj1) {
                        uint64_t e = as + jm;
                        if constexpr (IND) e = elist[e];
                        av_l = a_vals[e];
                        const uint64_t raw = chunk_off[e - (uint64_t)e0];
                        off_l = ((raw & kDirectBit) && (DIRECT || raw == kChunkSkip)) ? raw : raw - base + (uint64_t)pnl * W;  // (kChunkSkip and descriptors: as they are)
                    }
                    const uint32_t cj = (uint32_t)min((uint64_t)kWave, tj1 - jb + 1);
                    constexpr int QU = OSP_MUL_QU_HUB;   // chunks whose destinations are formed together (chunk_dests): with two entries per lane, twice as many chains
                    for (uint32_t q = 0; q < cj; q += QU) {
                        uint64_t off[QU];
                        T av[QU];
                        uint32_t lo[QU], hi[QU];
#pragma unroll
                        for (int i = 0; i < QU; i++) {
                            const bool there = q + i < cj;
                            const uint32_t qi = there ? q + i : q;
                            const uint64_t j = jb + qi;
                            av[i] = wave_bcast(av_l, qi);
                            off[i] = there ? wave_bcast(off_l, qi) : kChunkSkip;
                            lo[i] = j == tj0 ? tla : 0u;
                            hi[i] = off[i] == kChunkSkip ? 0u : (j == tj1 ? tlb : wp);   // (a skipped chunk: no lane in range)
                        }
#pragma unroll
                        for (uint32_t u = 0; u < PER; u++) {
                            const uint32_t lrel = u * kWave + lane;
                            Part<T> *dst[QU];
                            bool dir[QU];
                            chunk_dests<T, QU, MODE>(cells, hub, ri[u], chunk_off, stage, qstage, off, lrel, pnl * W + lrel, bc[u], dst, dir);
#pragma unroll
                            for (int i = 0; i < QU; i++)
                                if (lrel >= lo[i] && lrel < hi[i]) store_dest<T, MODE>(dst[i], off[i], dir[i], bc[u], av[i] * bv[u]);
                        }
                    }
                }
                x = pbeg + xe;
            }
        } else if (nb > 32) {
            // A entries in batches of 64: lane q fetches entry jb+q's value and chunk offset once, the inner loop
            // reads them with v_readlane -- no dependent global load per chunk
            for (uint64_t jb = j0; jb <= j1; jb += kWave) {
                const uint64_t jm = jb + lane;
                T av_l = 0;
                uint64_t off_l = kChunkSkip;
                if (jm <= j1) {
                    uint64_t e = as + jm;
                    if constexpr (IND) e = elist[e];
                    av_l = a_vals[e];
                    const uint64_t raw = chunk_off[e - (uint64_t)e0];
                    off_l = ((raw & kDirectBit) && (DIRECT || raw == kChunkSkip)) ? raw : raw - base;  // (kChunkSkip and descriptors: as they are)
                }
                const uint32_t cj = (uint32_t)min((uint64_t)kWave, j1 - jb + 1);
                for (uint32_t l0 = 0; l0 < nb; l0 += kWave) {
                    const uint32_t l = l0 + lane;
                    const bool in = l < nb;
                    uint32_t bc = 0, ri = 0; T bv = 0;
                    if (in) { bc = b_colidx[bs + l]; bv = b_vals[bs + l]; }
                    if constexpr (MODE == 2) { if (in) ri = hub.sx[bs + l + 1] - 1u - hub.sx[bs]; }   // the entry's run inside B's row (hub rows)
                    constexpr int QU = OSP_MUL_QU_MID;   // chunks whose destinations are formed together (chunk_dests)
                    for (uint32_t q = 0; q < cj; q += QU) {
                        uint64_t off[QU];
                        T av[QU];
                        bool ok[QU];
                        Part<T> *dst[QU];
#pragma unroll
                        for (int i = 0; i < QU; i++) {
                            const bool there = q + i < cj;
                            const uint32_t qi = there ? q + i : q;
                            const uint64_t j = jb + qi;
                            av[i] = wave_bcast(av_l, qi);
                            off[i] = there ? wave_bcast(off_l, qi) : kChunkSkip;
                            ok[i] = in && off[i] != kChunkSkip && !(j == j0 && l < la) && !(j == j1 && l >= lb);
                        }
                        bool dir[QU];
                        chunk_dests<T, QU, MODE>(cells, hub, ri, chunk_off, stage, qstage, off, l, l, bc, dst, dir);
#pragma unroll
                        for (int i = 0; i < QU; i++)
                            if (ok[i]) store_dest<T, MODE>(dst[i], off[i], dir[i], bc, av[i] * bv);
                    }
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
                    uint64_t e = as + j;
                    if constexpr (IND) e = elist[e];
                    const uint64_t raw = chunk_off[e - (uint64_t)e0];
                    if (raw != kChunkSkip) {
                        if (DIRECT && (raw & kDirectBit)) store_direct<T, MODE>(cells, hub, qstage, raw, bs, l, bc, a_vals[e] * bv);
                        else stream_store_part(&stage[raw - base + l], bc, a_vals[e] * bv);
                    }
                }
            }
        }
        cur = p0 + b;
    }
    if constexpr (!IND) return;
    }
}

// ---- long rows that are staged, row by row ----------------------------------------------------------------------------------
// With the short rows and the planned long rows gathered, what is left to stage are the long rows beyond the planner (a few
// hundred hub rows per panel in R-MAT-22: 4 % of the products, thousands of chunks each, most of them a handful of entries).
// Column by column -- multiply_kernel -- every such chunk costs a chain of dependent loads for its column (6 ms per launch for
// 1.4 GB of records).  This kernel walks the ROW instead: the chunk table holds its chunks in staging order, a wave takes a
// slice of the row's staging span, 64 chunks at a time (lane q: chunk q's span, B row and A value), and its lanes walk the
// products -- consecutive lanes write consecutive records: full lines, streamed.  Same products, same places (cscMulcsr,
// SimSpGEMM.cpp:265-281).
constexpr int kExpandThreads = 256;
constexpr uint32_t kExpandJob = 8192;   // products per workgroup
// jobs of long row h: those the multiply stages plainly (mode `split`, or `stretch` unless the panel's stretch rows are hub rows)
struct ExpandJobs {
    const uint32_t *rows; const uint64_t *row_off; const uint8_t *hmode; uint8_t mode_a, mode_b;
    __device__ uint64_t operator()(uint64_t h) const {
        const uint8_t m = hmode[h];
        if (m != mode_a && m != mode_b) return 0ull;
        const uint64_t U = row_off[rows[h] + 1] - row_off[rows[h]];
        return (U + kExpandJob - 1) / kExpandJob;
    }
};
template <class T>
__global__ __launch_bounds__(kExpandThreads) void expand_rows_kernel(
    const uint32_t *__restrict__ rows, uint32_t nlong, const uint64_t *__restrict__ jobbase, const uint64_t *__restrict__ row_off, uint64_t base,
    const uint32_t *__restrict__ rowfirst, const uint64_t *__restrict__ ct_off, const uint32_t *__restrict__ ct_bs, const uint32_t *__restrict__ perm,
    const T *__restrict__ a_vals, const uint32_t *__restrict__ b_colidx, const T *__restrict__ b_vals, Part<T> *__restrict__ stage) {
    const uint64_t job = blockIdx.x;
    if (job >= jobbase[nlong]) return;
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t h = (uint32_t)(upper_bound_dev(jobbase, 0, (uint64_t)nlong + 1, job) - 1);
    const uint32_t row = rows[h];
    const uint64_t rbeg = row_off[row], rend = row_off[row + 1];
    constexpr uint32_t per = kExpandJob / (kExpandThreads / kWave);
    const uint64_t p0 = rbeg + (job - jobbase[h]) * kExpandJob + (uint64_t)w * per;   // the wave's slice of the staging span
    if (p0 >= rend) return;
    const uint64_t p1 = min(p0 + per, rend);
    const uint32_t c0 = rowfirst[row], c1 = rowfirst[row + 1];
    // the chunk that holds product p0 (chunks without entries are never the answer: strictly ascending offsets decide)
    uint32_t c = (uint32_t)(upper_bound_dev(ct_off, (uint64_t)c0, (uint64_t)c1, p0) - 1);
    constexpr uint32_t kUnroll = 4;
    for (; c < c1; c += kWave) {
        const uint32_t cq = c + lane;
        const bool cv = cq < c1;
        const uint64_t o0 = cv ? ct_off[cq] : ~0ull, o1 = cv ? ct_off[cq + 1] : ~0ull;
        if (wave_bcast(o0, 0u) >= p1) break;   // (wave-uniform)
        const uint32_t bsv = cv ? ct_bs[cq] : 0u;
        const T av = cv ? (perm ? a_vals[perm[cq]] : a_vals[cq]) : T(0);   // (perm == nullptr: the values are in chunk order)
        // the group's products inside the slice
        const uint64_t g0 = max(wave_bcast(o0, 0u), p0);
        const uint32_t nval = min((uint32_t)kWave, c1 - c);
        const uint64_t g1 = min(wave_bcast(o1, nval - 1), p1);
        for (uint64_t ib = g0; ib < g1; ib += kUnroll * kWave) {
            uint32_t bc[kUnroll];
            T bv[kUnroll], cav[kUnroll];
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; u++) {
                const uint64_t i = ib + u * kWave + lane;
                uint32_t lo = 0, hi = kWave;   // last lane whose chunk starts at or before i (lanes past the end hold ~0: never)
#pragma unroll
                for (int step = 0; step < 6; step++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint64_t sv = (uint64_t)__shfl((long long)o0, (int)mid);
                    if (sv <= i) lo = mid; else hi = mid;
                }
                const uint64_t cst = (uint64_t)__shfl((long long)o0, (int)lo);
                const uint32_t cbs = (uint32_t)__shfl((int)bsv, (int)lo);
                cav[u] = __shfl(av, (int)lo);
                const uint32_t b = i < g1 ? cbs + (uint32_t)(i - cst) : 0u;   // clamped, branch-free: the loads stay in flight
                bc[u] = b_colidx[b];
                bv[u] = b_vals[b];
            }
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; u++) {
                const uint64_t i = ib + u * kWave + lane;
                if (i < g1) stream_store_part(&stage[i - base], bc[u], cav[u] * bv[u]);
            }
        }
    }
}

// ---- tile planning -----------------------------------------------------------------------------
// Tiles are consecutive rows packed greedily up to the tile capacity (and kTileMaxRows rows); a row
// longer than the capacity is a tile of its own.  Greedy packing is sequential, so the rows are first
// cut into coarse blocks of ~8 tiles at fixed staging offsets; one thread walks each block.
struct CoarseStartFlag {
    const uint64_t *row_off;
    uint64_t r0, base, slot;
    const uint8_t *force;  // optional: rows that must start a tile (first segment of a split long row)
    __device__ uint32_t operator()(uint64_t t) const {
        if (t == 0) return 1;
        const uint64_t r = r0 + t;
        if (force && force[t]) return 1;
        return ((row_off[r] - base) / slot != (row_off[r - 1] - base) / slot || (t & 0xffffu) == 0) ? 1u : 0u;
    }
};
// counts[j] = tiles of coarse block j (tile_rows == nullptr), or write them at tile_base[j]
// (The number of coarse blocks stays on the device -- *ncb_p; the launch covers the host's upper bound ncb_max, and the
// blocks beyond the real count get 0 tiles -- so that planning the tiles costs one read-back, not two.)
__global__ void tile_walk_kernel(const uint32_t *cb_rows, const uint32_t *ncb_p, uint32_t ncb_max, uint64_t r_end, const uint64_t *row_off,
                                 uint64_t cap, uint32_t max_rows, const uint32_t *tile_base, uint32_t *counts,
                                 uint32_t *tile_rows) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ncb = *ncb_p;
    if (j >= ncb) {
        if (!tile_rows && j < ncb_max) counts[j] = 0;
        return;
    }
    uint64_t r = cb_rows[j];
    const uint64_t end = (j + 1 < ncb) ? (uint64_t)cb_rows[j + 1] : r_end;
    uint32_t n = 0;
    const uint32_t out = tile_rows ? tile_base[j] : 0u;
    while (r < end) {
        if (tile_rows) tile_rows[out + n] = (uint32_t)r;
        const uint64_t lim = min(end, r + max_rows);
        // last r2 in (r, lim] whose rows [r, r2) still fit
        uint64_t r2 = upper_bound_dev(row_off, r + 1, lim + 1, row_off[r] + cap) - 1;
        if (r2 <= r) r2 = r + 1;  // a row longer than a tile stands alone
        r = r2;
        n++;
    }
    if (!tile_rows) counts[j] = n;
}
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
// zeros).  One workgroup per tile; tiles are handed out in output order by a ticket counter and a
// decoupled look-back over `tile_status` gives every tile its offset in the final CSR, so merged
// rows are written ONCE, straight to c_col / c_val / c_rowptr (no per-row compaction pass).
constexpr uint64_t kStatusAgg = 1ull << 62, kStatusPrefix = 2ull << 62, kStatusMask = (1ull << 62) - 1;

// (8 until late in round 3.  With the runs' first entries summed from registers, 16 costs the skewed products nothing and a
// product whose output rows are dense -- 32768^2, 634 entries per row, every output entry fed by ~12 products -- a third less
// merge time: 57.2 -> 40.2 ms per launch; 32 measures the same as 16.)
#ifndef OSP_RUN_SHORT
#define OSP_RUN_SHORT 16
#endif
constexpr int kRunShort = OSP_RUN_SHORT;  // entries of a run its head thread sums itself; the rest of a longer run: a wave (merge_tiles_kernel)
template <class T> __device__ __forceinline__ T index_as_value(uint32_t i);
template <> __device__ __forceinline__ float index_as_value<float>(uint32_t i) { return __uint_as_float(i); }
template <> __device__ __forceinline__ double index_as_value<double>(uint32_t i) { return __longlong_as_double((long long)i); }
__device__ __forceinline__ uint32_t value_as_index_impl(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t value_as_index_impl(double v) { return (uint32_t)__double_as_longlong(v); }
template <class T> __device__ __forceinline__ uint32_t value_as_index(T v) { return value_as_index_impl(v); }
#ifndef OSP_DIGIT_BITS
#define OSP_DIGIT_BITS 10
#endif
constexpr int kDigitBits = OSP_DIGIT_BITS;    // widest radix of one LDS sort pass: 20 key bits sort in two passes
constexpr int kDigits = 1 << kDigitBits;      // 1024 buckets

// LDS of one merge workgroup: 14 bytes per tile entry + the digit counters.
//   key0 | pos0 | pad | key1 | pos1     two (key, staging position) buffers the sort passes ping-pong between
// The values never take part in the sort: they wait in registers and are written, after the last pass, into
// the 8 bytes per entry that do not hold the sorted keys -- (key0, pos0, pad) when the sorted keys sit in
// buffer 1, (pad, key1, pos1) when they sit in buffer 0; that is what `pad` is for.  Before the first pass the
// hash set that counts the tile's distinct keys uses everything behind key0 (10 bytes = 2.5 words per entry).
// f32: values are 4 bytes and fit the idle KEY buffer alone, so `pad` shrinks to nothing (12 bytes per entry, hash table 2 words).
// After the last pass the digit counters are dead and hold `rank` (output slot per sorted position).
template <class T, int NT, int CAP = TileCap<T>::value>
struct alignas(8) MergeSmem {
    static constexpr int kTileCap = CAP;
    static_assert(CAP % 4 == 0, "the value area behind key0/pos0 must stay 8-byte aligned");
    static_assert(sizeof(T) <= 8, "values alias 8 bytes per entry");
    static constexpr bool kWide = sizeof(T) == 8;   // f64 values need the 2 padding bytes per entry, f32 values fit an idle key buffer
#ifndef OSP_HASH_IN_CNT
#define OSP_HASH_IN_CNT 1
#endif
    // words behind key0, and the digit counters behind them (idle until the first pass zeroes them): the emptier the table,
    // the fewer probes (load factor 0.40 -> 0.26 for f64 tiles)
    static constexpr uint32_t kHashWords = (kWide ? (uint32_t)CAP * 5u / 2u : (uint32_t)CAP * 2u) +
                                           (OSP_HASH_IN_CNT ? (kWide ? 0u : 2u) + (uint32_t)(NT / kWave) * (1u << OSP_DIGIT_BITS) / 2u : 0u);
    uint32_t key0[CAP];
    uint16_t pos0[CAP];
    uint16_t pad[kWide ? CAP : 4];
    uint32_t key1[CAP];
    uint16_t pos1[CAP];
    union {
        uint16_t cnt[NT / kWave][kDigits];
        uint16_t rank[CAP + 1];
    };
    uint32_t rowo[kTileMaxRows + 1];
    uint32_t gbits[CAP / 32];   // gathered tiles (merge_tiles_kernel): one bit per entry that starts a run; all clear between tiles
    uint32_t scratch[NT / kWave + 1];
    uint32_t nlongrun;  // runs longer than kRunShort, summed by whole waves (their positions and sums: behind `rank`)
    uint32_t hcount;  // distinct (row, col) keys of the tile, counted by hashing before the sort
    uint64_t excl;
    __device__ __forceinline__ uint32_t *key(int c) { return c ? key1 : key0; }
    __device__ __forceinline__ uint16_t *pos(int c) { return c ? pos1 : pos0; }
    // where the values go once the sorted keys sit in buffer `c`
    __device__ __forceinline__ T *vals(int c) {
        return reinterpret_cast<T *>(c ? reinterpret_cast<char *>(key0) : kWide ? reinterpret_cast<char *>(pad) : reinterpret_cast<char *>(key1));
    }
    __device__ __forceinline__ uint32_t *htab() { return reinterpret_cast<uint32_t *>(pos0); }
    // long runs: the part of the dead digit counters that `rank` leaves free
    static constexpr int kMaxLong = CAP / kRunShort;
    static constexpr size_t kLongOff = ((CAP + 1) * sizeof(uint16_t) + 7) / 8 * 8;
    static_assert(kLongOff + kMaxLong * (sizeof(T) + sizeof(uint16_t)) <= sizeof(uint16_t) * (NT / kWave) * kDigits, "long-run list: behind rank, inside the counters");
    __device__ __forceinline__ T *long_sum() { return reinterpret_cast<T *>(reinterpret_cast<char *>(cnt) + kLongOff); }
    __device__ __forceinline__ uint16_t *long_pos() { return reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(cnt) + kLongOff + kMaxLong * sizeof(T)); }
};

// Rank of this lane among the lanes of its wave that hold the same digit (lower lanes first), and
// the size of that group.  BITS ballots; per bit one sign mask, one xnor and one and per half.
template <int BITS>
__device__ __forceinline__ void wave_match_digit(unsigned digit, bool valid, unsigned &rank, unsigned &count) {
    const uint64_t vm = __ballot(valid);
    uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const uint32_t bit = (digit >> b) & 1u;
        const uint64_t m = __ballot(bit != 0);
        const uint32_t sbm = 0u - bit;  // all ones where my bit is set
        plo &= ~((uint32_t)m ^ sbm);
        phi &= ~((uint32_t)(m >> 32) ^ sbm);
    }
    rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    count = __popc(plo) + __popc(phi);
}

// Self-test run once per context: does one LDS atomic-add instruction hand out its old values in ascending lane order
// among the lanes that hit the same counter?  The stable ranks of every radix pass rely on it (OSP_RANK_ATOMIC); it holds
// on gfx950 but is not documented, so a device where it does not must fail loudly rather than sum in another order.
__global__ __launch_bounds__(256) void rank_order_selftest_kernel(uint32_t *bad) {
    __shared__ uint32_t cnt[4][64];
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    uint32_t x = 0x9E3779B9u * (blockIdx.x * 256u + tid + 1u), nbad = 0;
    for (int r = 0; r < 64; r++) {
        cnt[w][lane] = 0;
        for (int it = 0; it < 4; it++) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const unsigned dg = (x >> 9) & ((r & 1) ? 3u : 127u);  // every other round: four counters for 64 lanes
            const bool valid = ((x >> 27) & 7u) != 0;
            unsigned rk, c;
            wave_match_digit<7>(dg, valid, rk, c);
            const unsigned half = 16u * (dg & 1u);
            const uint32_t before = (cnt[w][dg >> 1] >> half) & 0xffffu;
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const uint32_t got = (atomicAdd(&cnt[w][dg >> 1], 1u << half) >> half) & 0xffffu;
                nbad += got != before + rk;
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

// (Tickets: the persistent workgroups of merge_tiles_kernel take their tiles from ONE counter word, and that word was what
// bounded the kernel in rounds 3-4 (staged tiles: 77 per microsecond); since round 5 a tile also forms its partial products and the
// kernel takes 73 per microsecond for what its workgroups compute -- but the word is the next ceiling: a tile more than a tenth
// cheaper will not show before two tiles share a ticket.  A word serves about 88 returning atomics per microsecond on MI355X whoever asks
// (/opt/skills/guides/MI355X_MICROARCH.md, price list, "dequeue"); the kernel hands out 68-76 tiles per microsecond, and
// tools/bench_merge with the sort AND the look-back switched off still takes 2.19 ms for 174 763 tiles (80 per
// microsecond) -- while the same kernel with tiles assigned statically (no ticket, no look-back) SORTS them in 1.79 ms, and
// loads, hashes and stores them in 1.31.  What was tried to lift the bound, all measured on those tiles (2.40 ms as is):
//   * static assignment with the look-back: 3.03 ms -- a slow workgroup stalls everything behind it;
//   * a two-level dispenser -- eight group words holding chunks of 32 tickets claimed on demand from a global counter:
//     correct for every grid size, 4.0 ms -- 160 workgroups per group drain a chunk faster than its refill's two dependent
//     atomics come back;
//   * the same in-order sequence spread over 4 or 8 words (word w hands out tiles w, w+W, ...; a workgroup reads all the
//     words and increments the one whose next tile is smallest, which keeps the guarantee that the smallest tile nobody
//     holds is what the next free workgroup takes): correct for every grid size, 6.3 / 5.8 ms -- the reads of the hot
//     counter lines cost what the atomics cost, and there are W of them per ticket;
//   * words bound to groups of workgroups (no reads) would lift it, but a tile could then wait for a tile that only a
//     workgroup not yet resident can take: with two processes on one device that is a deadlock; not done.
// What is left is fewer, larger tiles -- LDS per entry is what limits that (five workgroups of 1536 entries per CU beat
// four of 2048, see TileCap).)
// Exclusive prefix of tile `t` by decoupled look-back (called by wave 0 of the block).
// One round trip fetches kLookWin windows of 64 predecessors.  (Measured: wide windows lose -- the
// extra polling traffic costs more than the walk saves -- so kLookWin = 1; what matters is that tiles
// are ticketed in the order they will finish, see merge_tiles_kernel.)
constexpr int kLookWin = 1;
// Make tile t's entry count visible to its successors (one lane).
__device__ __forceinline__ void lookback_publish(uint64_t *status, uint32_t t, uint64_t total) {
    __hip_atomic_store(&status[t], (t == 0 ? kStatusPrefix : kStatusAgg) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Tickets in SHARDS (merge_tiles_kernel, nshards > 1).  One counter word serves ~88 returning atomics per microsecond
// whoever asks, and the tile kernel asks for one per tile: with the look-back and the hash count switched off the kernel
// still takes 2.19 ms for 174 763 tiles, against 1.76 ms with tiles assigned statically (tools/bench_merge).  With S shards,
// tile t belongs to shard t mod S and every shard hands out ITS tiles in ascending order from a counter of its own (one
// 128-byte line each).  A workgroup joins the shard its ARRIVAL number selects (one atomic per workgroup on a further
// word: the workgroups that are running cover the shards round-robin in the order they started, whatever the dispatcher
// does) and moves on to the next shard when its own is exhausted.
// What it keeps of the single counter's guarantee -- a tile only waits for tiles that running workgroups hold -- : the
// smallest unfinished tile is the next ticket of its shard, so some workgroup of that shard that is running (or starts)
// takes it; that needs at least S workgroups of the launch to get to run side by side at some time, where the single
// counter needs one.  The look-back therefore gives up after kLookbackSpinLimit fruitless rounds (seconds), raises the
// product's abort word, and the host reports an error instead of a hang; (the library launches with one counter).
constexpr int kTicketStride = 32;                    // 32-bit words between two shard counters
constexpr uint32_t kLookbackSpinLimit = 1u << 22;
__device__ __forceinline__ uint32_t take_ticket(uint32_t *ticket, uint32_t nshards, uint32_t &shard, uint32_t ntiles) {
    if (nshards <= 1) return atomicAdd(ticket, 1u);
    for (uint32_t tries = 0; tries < nshards; tries++) {
        const uint32_t n = atomicAdd(&ticket[shard * kTicketStride], 1u);
        const uint64_t t = (uint64_t)n * nshards + shard;
        if (t < ntiles) return (uint32_t)t;
        shard = shard + 1 == nshards ? 0u : shard + 1;   // this shard's tiles are gone: help the next one
    }
    return ntiles;
}
template <bool PUBLISH = true>
__device__ __forceinline__ uint64_t lookback_prefix(uint64_t *status, uint32_t t, uint64_t total, uint32_t *abort_word = nullptr) {
    const unsigned lane = lane_id();
    if (PUBLISH && lane == 0) lookback_publish(status, t, total);
    uint64_t excl = 0;
    int64_t b = (int64_t)t - 1;
    uint32_t spins = 0;
    while (b >= 0) {
        uint64_t sv[kLookWin];
#pragma unroll
        for (int j = 0; j < kLookWin; j++) {
            const int64_t idx = b - (int64_t)(j * kWave) - (int64_t)lane;
            sv[j] = kStatusPrefix;  // virtual tiles before 0: prefix 0
            if (idx >= 0) sv[j] = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool done = false, retry = false;
        uint64_t part = 0;
#pragma unroll
        for (int j = 0; j < kLookWin; j++) {
            if (!done && !retry) {
                const uint64_t m_prefix = __ballot((sv[j] >> 62) == 2), m_empty = __ballot((sv[j] >> 62) == 0);
                const int p = m_prefix ? __builtin_ctzll(m_prefix) : 64;  // nearest predecessor holding a prefix
                const uint64_t need = p >= 63 ? ~0ull : ((2ull << p) - 1ull);
                if (m_empty & need) {
                    retry = true;  // someone nearer has not published yet
                } else {
                    part += ((int)lane <= p) ? (sv[j] & kStatusMask) : 0ull;
                    if (p < 64) done = true;
                }
            }
        }
        if (retry) {
            __builtin_amdgcn_s_sleep(1);
            if (abort_word && (++spins & 255u) == 0) {   // (wave-uniform)
                // nobody is going to publish what this tile waits for, or somebody else found that out: stop waiting (the
                // partial prefix keeps every write inside the output; the host turns the word into an error)
                if (spins >= kLookbackSpinLimit && lane == 0) atomicExch(abort_word, 1u);
                if (spins >= kLookbackSpinLimit || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            }
            continue;  // keep what is already summed? no: re-read all
        }
        excl += wave_reduce_sum_u62(part);
        if (done) break;
        b -= (int64_t)kLookWin * kWave;
    }
    if (lane == 0 && t != 0)
        __hip_atomic_store(&status[t], kStatusPrefix | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// One merge tile: rows [ra, ra+nr) of level `lvl` whose partial products are stage[lvl][s, s+n).
// Level 0 = output rows in the staging buffer; level 1 = column-range segments of the split long rows
// in the second buffer.  All tiles of a panel form ONE chain in output order: a long row's placeholder
// in the level-0 list is replaced by the tiles of its segments.
struct TileDesc {
    uint64_t s;
    uint32_t ra, nr, n, lvl;
    // sort key of a level-1 tile: its segments are consecutive column ranges of ONE long row, so the column
    // itself orders them; key = col - cbase needs kbits bits (about log2(N * tile / row length), often <= 20:
    // two sort passes).  kbits = 0: level-0 key (local row, col).
    uint32_t cbase, kbits;
    // a tile of a GATHERED row (below): its runs are descriptors [rbeg, rbeg + rcnt) of the panel's run table, nothing of
    // it was staged; rcnt = 0: the tile's records are in the staging buffer of its level
    uint32_t rbeg, rcnt;
};
// ---- gathered rows (round 5) ---------------------------------------------------------------------------------------------
// A direct row's plan (direct_plan_kernel, osp_split.h) knows, for every chunk (non-zero A[i,k]) and column range of the row,
// which entries of B's row k fall into the range: a RUN of consecutive entries, because B's rows are sorted.  Until round 4 the
// multiply wrote every run to its place in the second buffer and the tile kernel read it back: 2 x 12 bytes of HBM traffic per
// partial product, the writes in runs of 8-75 records at unaligned addresses (the multiply's bound).  A GATHERED row is never
// written: the planner leaves one descriptor per non-empty run -- where the run would have started in the second buffer
// (dst: that buffer stays virtual), where it starts in B's arrays (src) and the chunk's value of A -- in the order
// (row, range, chunk), and the tile of a range forms its products itself: a record at virtual position p of the tile belongs
// to the last run with dst <= p, it is entry src + (p - dst) of B times av.  Same records in the same order as the written
// tile, so the sort and the sums are bit-identical; the multiply skips the row's chunks (kChunkSkip).
// cscMulcsr (SimSpGEMM.cpp:265-281) for these rows therefore happens inside merge_tiles_kernel.
constexpr uint32_t kNoRuns = 0xffffffffu;   // vrun_off of a segment whose records are in the second buffer
template <class T> struct RunDesc;
template <> struct alignas(16) RunDesc<double> { uint32_t dst, src; double av; };
template <> struct alignas(4) RunDesc<float> { uint32_t dst, src; float av; };
// SHORT rows (those that fit a tile) are gathered the same way: their runs are their chunks -- the chunk table of the symbolic
// phase as descriptors (dst: the chunk's staging offset, which stays virtual), chunks without entries left out, made once per
// product (short_runs_kernel); a tile of rows [ra, ra + nr) takes descriptors [rowfirst0[ra], rowfirst0[ra + nr)).  With both,
// only rows beyond the planner (hub rows, rows with a range that exceeds a tile) are still multiplied into HBM.
template <class T>
struct GatherArgs {
    const RunDesc<T> *runs = nullptr;     // the panel's run table (null: no gathered long rows)
    const RunDesc<T> *runs0 = nullptr;    // the product's short rows (null: they are staged)
    const uint32_t *b_colidx = nullptr;
    const T *b_vals = nullptr;
};
// flag(c) = chunk c (in (row, k) order) has entries.  (The chunks of long rows get descriptors too -- nobody reads them: telling
// them apart costs two gathers from the row offsets per chunk, twice, and the table is allocated for every chunk anyway.)
struct ShortRunFlag {
    const uint64_t *off;
    __device__ uint32_t operator()(uint64_t c) const { return off[c + 1] > off[c] ? 1u : 0u; }
};
template <class T>
__global__ void short_runs_kernel(ShortRunFlag f, const uint32_t *cidx, uint64_t nnz, const uint32_t *bs, const T *av_sorted, RunDesc<T> *runs0) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nnz || !f(c)) return;
    RunDesc<T> rd;
    rd.dst = (uint32_t)f.off[c];
    rd.src = bs[c];
    rd.av = av_sorted[c];
    runs0[cidx[c]] = rd;
}
__global__ void short_rowfirst_kernel(const uint32_t *rowfirst, const uint32_t *cidx, uint64_t M, uint32_t *rowfirst0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= M) rowfirst0[i] = cidx[rowfirst[i]];
}
// A gathered row may have a range that exceeds a tile (the hub columns: one column fed by more chunks than a tile holds
// entries).  Such a segment goes to the paths for over-long segments -- dense accumulators, big in-place tiles, the global sort
// -- which read RECORDS: this kernel writes the segment's records where they would have been written (its place in the second
// buffer), from its run descriptors; everything else of the row stays virtual.  Jobs of kExpandJob products, a wave per slice
// (a segment of a row of many chunks holds thousands of runs: the slice's first run is found with 64 probes a step).
// jobs of over-long segment number t of the list: kExpandJob products each; segments of rows that are not gathered have none
struct SegExpandJobs {
    const uint32_t *seglist; const uint64_t *vrow_off; const uint32_t *vrun_off;
    __device__ uint64_t operator()(uint64_t t) const {
        const uint32_t v = seglist[t];
        if (vrun_off[v] == kNoRuns) return 0ull;
        return (vrow_off[v + 1] - vrow_off[v] + kExpandJob - 1) / kExpandJob;
    }
};
template <class T>
__global__ __launch_bounds__(kExpandThreads) void expand_segments_kernel(
    const uint32_t *__restrict__ seglist, uint32_t nseg, const uint64_t *__restrict__ jobbase, const uint64_t *__restrict__ vrow_off,
    const uint32_t *__restrict__ vrun_off, const uint32_t *__restrict__ vrun_end, const RunDesc<T> *__restrict__ runs,
    const uint32_t *__restrict__ b_colidx, const T *__restrict__ b_vals, Part<T> *__restrict__ qstage) {
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
    constexpr uint32_t NW = kExpandThreads / kWave, per = kExpandJob / NW, kUnroll = 4;
    const uint64_t njobs = jobbase[nseg];
    // (a fixed grid strides over the jobs: the host knows a bound of their number only -- every long row's products)
    for (uint64_t job = blockIdx.x; job < njobs; job += gridDim.x) {
        const uint32_t t = (uint32_t)(upper_bound_dev(jobbase, 0, (uint64_t)nseg + 1, job) - 1);
        const uint32_t v = seglist[t];
        const uint32_t r0 = vrun_off[v], R = vrun_end[v] - r0;
        const uint64_t s = vrow_off[v], n = vrow_off[v + 1] - s;
        const uint32_t s32 = (uint32_t)s;   // (run positions are 32 bits wide, modulo)
        const RunDesc<T> *__restrict__ rd = runs + r0;
        const uint64_t p0 = (job - jobbase[t]) * kExpandJob + (uint64_t)w * per;   // the wave's slice of the segment
        if (p0 >= n) continue;
        const uint64_t p1 = min(p0 + per, n);
        // last run that starts at or before p0: 64 probes a step (a segment of a hub row holds tens of thousands of runs)
        uint32_t lo = 0, hi = R;   // the answer is in [lo, hi)
        while (hi - lo > 1) {
            const uint32_t span = hi - lo, step = (span + kWave - 1) / kWave;
            const uint32_t q = lo + lane * step;                         // probes lo, lo + step, ... (those below hi count)
            const bool le = q < hi && (uint64_t)(rd[q].dst - s32) <= p0;  // monotone: a prefix of the lanes
            const uint32_t c = (uint32_t)__popcll(__ballot(le));          // >= 1: the run at lo starts at or before p0
            const uint32_t nlo = lo + (c - 1) * step;
            hi = min(hi, nlo + step);
            lo = nlo;
        }
        for (uint32_t c = lo; c < R; c += kWave) {
            const uint32_t cq = c + lane;
            const bool cv = cq < R;
            const RunDesc<T> d = rd[cv ? cq : c];
            const uint64_t o0 = cv ? (uint64_t)(d.dst - s32) : ~0ull;
            if (wave_bcast(o0, 0u) >= p1) break;   // (wave-uniform)
            const uint32_t nval = min((uint32_t)kWave, R - c);
            // where the group ends: the start of the run behind it, or the segment's end
            uint64_t gend = n;
            if (c + nval < R) gend = (uint64_t)(rd[c + nval].dst - s32);
            const uint64_t g0 = max(wave_bcast(o0, 0u), p0), g1 = min(gend, p1);
            for (uint64_t ib = g0; ib < g1; ib += kUnroll * kWave) {
                uint32_t bc[kUnroll];
                T bv[kUnroll], cav[kUnroll];
#pragma unroll
                for (uint32_t u = 0; u < kUnroll; u++) {
                    const uint64_t i = ib + u * kWave + lane;
                    uint32_t llo = 0, lhi = kWave;   // last lane whose run starts at or before i
#pragma unroll
                    for (int step = 0; step < 6; step++) {
                        const uint32_t mid = (llo + lhi) >> 1;
                        const uint64_t sv = (uint64_t)__shfl((long long)o0, (int)mid);
                        if (sv <= i) llo = mid; else lhi = mid;
                    }
                    const uint64_t cst = (uint64_t)__shfl((long long)o0, (int)llo);
                    const uint32_t cbs = (uint32_t)__shfl((int)d.src, (int)llo);
                    cav[u] = __shfl(d.av, (int)llo);
                    const uint32_t b = i < g1 ? cbs + (uint32_t)(i - cst) : 0u;   // clamped, branch-free
                    bc[u] = b_colidx[b];
                    bv[u] = b_vals[b];
                }
#pragma unroll
                for (uint32_t u = 0; u < kUnroll; u++) {
                    const uint64_t i = ib + u * kWave + lane;
                    if (i < g1) qstage[s + i] = Part<T>{bc[u], cav[u] * bv[u]};
                }
            }
        }
    }
}

// Everything merge_tiles_kernel needs per level.
template <class T>
struct MergeLevels {
    const Part<T> *stage[2];
    const uint64_t *row_off[2];
    uint64_t base[2];
    int64_t *c_rowptr[2];          // [0] the final rowptr, [1] offsets of the segments (same output space)
    const uint32_t *heavy_nnz[2];  // entry counts of rows reduced outside the tile kernel (global-sort path)
};
// chain position of the level-`lvl` tile list; for level 0: j0 = sorted placeholder indices of the long
// rows, extra = exclusive scan of (segment tiles of that long row - 1)
template <int CAP>
__global__ void tile_desc_kernel(const uint32_t *tile_rows, uint32_t ntiles, uint64_t r_end, const uint64_t *row_off,
                                 uint64_t base, uint32_t lvl, const uint32_t *j0, const uint32_t *extra, uint32_t nlong,
                                 const uint32_t *tb, const uint32_t *vcol0, const uint32_t *vcol1, TileDesc *desc,
                                 const uint32_t *vrun_off = nullptr, const uint32_t *vrun_end = nullptr, const uint32_t *rowfirst0 = nullptr) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const uint64_t ra = tile_rows[t], rb = (t + 1 < ntiles) ? (uint64_t)tile_rows[t + 1] : r_end;
    TileDesc d;
    d.s = row_off[ra] - base;
    d.ra = (uint32_t)ra;
    d.nr = (uint32_t)(rb - ra);
    d.n = (uint32_t)min(row_off[rb] - base - d.s, (uint64_t)CAP + 1);  // CAP+1 = "a single long row"
    d.lvl = lvl;
    d.cbase = 0;
    d.kbits = 0;
    d.rbeg = 0;
    d.rcnt = 0;
    if (lvl == 0 && rowfirst0 && d.n <= (uint32_t)CAP) {   // short rows, gathered: their chunks are their runs
        d.rbeg = rowfirst0[ra];
        d.rcnt = rowfirst0[rb] - d.rbeg;
    }
    uint32_t pos = t;
    if (nlong) {
        if (lvl == 0) {
            if (d.n > (uint32_t)CAP) return;  // placeholder of a split long row: its segment tiles stand here
            const uint32_t h = (uint32_t)lower_bound_dev(j0, 0, (uint64_t)nlong, t);  // long rows before tile t
            pos = t + extra[h];
        } else {
            const uint32_t h = (uint32_t)(upper_bound_dev(tb, 0, (uint64_t)nlong + 1, t) - 1);  // owner long row
            pos = j0[h] + extra[h] + (t - tb[h]);
            // the tile's segments are consecutive column ranges of one long row: [vcol0[ra], vcol1[rb - 1])
            d.cbase = vcol0[ra];
            const uint64_t range = (uint64_t)(vcol1[rb - 1] - vcol0[ra]);
            int kb = 1;
            while (kb < 32 && (1ull << kb) < range) kb++;
            d.kbits = (uint32_t)kb;
            // segments of a gathered row: their runs are consecutive in the run table (kNoRuns: the row was written or split)
            if (vrun_off && vrun_off[ra] != kNoRuns && d.n <= (uint32_t)CAP) {
                d.rbeg = vrun_off[ra];
                d.rcnt = vrun_end[rb - 1] - d.rbeg;
            }
        }
    }
    desc[pos] = d;
}
// per split long row h: placeholder index in the level-0 tile list, first level-1 tile, tile count - 1
__global__ void chain_rows_kernel(const uint32_t *heavy_rows, uint32_t nlong, const uint32_t *tile_rows0, uint32_t ntiles0,
                                  const uint64_t *vbase, const uint32_t *tile_rows1, uint32_t ntiles1, uint32_t *j0,
                                  uint32_t *tb, uint32_t *cntm1) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > nlong) return;
    if (h == nlong) { tb[h] = ntiles1; return; }
    j0[h] = (uint32_t)lower_bound_dev(tile_rows0, 0, (uint64_t)ntiles0, heavy_rows[h]);
    const uint32_t b = (uint32_t)lower_bound_dev(tile_rows1, 0, (uint64_t)ntiles1, vbase[h]);
    const uint32_t e = (uint32_t)lower_bound_dev(tile_rows1, 0, (uint64_t)ntiles1, vbase[h + 1]);
    tb[h] = b;
    cntm1[h] = e - b - 1;
}
// after the merge: rowptr of the split long rows (= offset of their first segment) and of the panel end
__global__ void chain_finish_kernel(const uint32_t *heavy_rows, uint32_t nlong, const uint64_t *vbase, const int64_t *vptr,
                                    const uint64_t *out_end, uint64_t r_end, int64_t *c_rowptr) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < nlong) c_rowptr[heavy_rows[h]] = vptr[vbase[h]];
    if (h == 0) c_rowptr[r_end] = (int64_t)*out_end;
}

// Debug build (`make debug`, -DOSP_CHECK_DESC; compiled out of the library otherwise).  Every thread compares the tile
// descriptor it got through LDS with the one in global memory, and every workgroup leaves breadcrumbs -- phase, tile
// and the descriptor it is working on -- in PINNED HOST memory, which the host can still read after a device fault
// (dbg_sync prints them under OSP_SYNC=1).  This is what found round 1's intermittent fault: workgroups at "kernel
// start" holding tile numbers beyond the tile count and garbage descriptors.
#ifdef OSP_CHECK_DESC
__device__ unsigned long long osp_desc_bad[16];
#define OSP_DESC_CHECK(t_, d_) do { if ((t_) < ntiles) { const TileDesc g_ = desc[(t_)]; \
    if (g_.s != (d_).s || g_.n != (d_).n || g_.ra != (d_).ra || g_.nr != (d_).nr || g_.lvl != (d_).lvl) { \
        if (atomicAdd(&osp_desc_bad[0], 1ull) == 0) { osp_desc_bad[1] = (t_); osp_desc_bad[2] = tid; osp_desc_bad[3] = (d_).s; osp_desc_bad[4] = g_.s; \
            osp_desc_bad[5] = (d_).n; osp_desc_bad[6] = g_.n; osp_desc_bad[7] = (d_).lvl; osp_desc_bad[8] = blockIdx.x; osp_desc_bad[9] = NT; } \
        (d_) = g_; } } } while (0)
// breadcrumbs in pinned host memory (readable after a device fault): what every workgroup was doing last
__device__ unsigned long long *osp_crumbs;
#define OSP_CRUMB(ph_, a_, b_, c_) do { if (tid == 0 && osp_crumbs) { volatile unsigned long long *p_ = osp_crumbs + (size_t)(blockIdx.x & 4095u) * 8u; \
    p_[1] = t; p_[2] = d.s; p_[3] = d.n; p_[4] = (a_); p_[5] = (b_); p_[6] = (c_); p_[7] = d.ra; \
    p_[0] = (unsigned long long)(ph_) | ((unsigned long long)NT << 8) | ((unsigned long long)ABL << 32); __threadfence_system(); } } while (0)
#else
#define OSP_DESC_CHECK(t_, d_)
#define OSP_CRUMB(ph_, a_, b_, c_)
#endif

// Phase timing for tools/bench_merge.hip (-DOSP_MERGE_PROF): thread 0 of every workgroup adds the cycles between
// marks to osp_merge_prof[phase].  Compiled out of the library.
#ifdef OSP_MERGE_PROF
__device__ unsigned long long osp_merge_prof[16];
#define OSP_PROF_DECL __shared__ unsigned long long prof_acc[12]; if (tid < 12) prof_acc[tid] = 0; __syncthreads(); unsigned long long prof_t = clock64();
#define OSP_PROF_MARK(k) do { if (tid == 0) { const unsigned long long now_ = clock64(); prof_acc[k] += now_ - prof_t; prof_t = now_; } } while (0)
#define OSP_PROF_FLUSH do { if (tid == 0) for (int k_ = 0; k_ < 12; k_++) atomicAdd(&osp_merge_prof[k_], prof_acc[k_]); } while (0)
#else
#define OSP_PROF_DECL
#define OSP_PROF_MARK(k)
#define OSP_PROF_FLUSH
#endif

// ABL: ablation switches for tools/bench_merge.hip only (1 = no sort, 2 = no look-back, 4 = no ticket,
// 8 = entry count published after the sort instead of by hashing, 128 / 256 = narrow LDS accesses for the digit counters /
// the hash table).
// Bit 32 is a real mode, not an ablation: IN-PLACE tiles -- every tile is one over-long segment that is reduced where
// it lies (records written back over its own beginning, entry count to heavy_nnz); no offset chain, no look-back.
// the library always instantiates ABL = 0.
//
// Persistent workgroups: each takes tiles from the ticket counter until none are left.  The ticket and
// the descriptor of the NEXT tile are fetched by thread 0 while the current tile is being merged, and a
// tile's row offsets and partial products are requested together, so one tile costs one exposed HBM
// round trip instead of a chain of five.
// workgroups of NT threads that fit one CU's 160 KiB of LDS -> waves per SIMD the register budget must allow
constexpr int kMergeMaxWgs = 5;  // per CU (measured: five workgroups at <= 102 registers, a few spilled, beat four at 120)
template <class T, int NT, int CAP, int MAXWG = kMergeMaxWgs, int ABL = 0>
constexpr int merge_wgs_per_cu() {
    // (mode 64, row-wise tiles, keeps the values of a tile in LDS: one more array)
    const int wgs = (160 * 1024) / (int)(sizeof(MergeSmem<T, NT, CAP>) + ((ABL & 64) ? CAP * sizeof(T) : 0) + 64);
    return wgs > MAXWG ? MAXWG : wgs;
}
template <class T, int NT, int CAP, int MAXWG = kMergeMaxWgs, int ABL = 0>
constexpr int merge_waves_per_simd() {
    const int w = (merge_wgs_per_cu<T, NT, CAP, MAXWG, ABL>() * NT + 255) / 256;  // (rounded up: 3 workgroups of 6 waves need 5 per SIMD)
    return w > 8 ? 8 : (w < 1 ? 1 : w);
}
// RA: stable ranks from the return order of one LDS atomic (true) or from ballot matching (false); a context whose
// self-test of that order fails runs the ballot instantiations (osp_api.hip, Context::rank_atomic).
// GA: the launch has tiles of gathered rows (TileDesc::rcnt != 0; `ga`: their run table and B); launches without them run the
// instantiation that does not know them (a gathered tile keeps its values in registers through the sort: 12 more for f64)
template <class T, int NT, int ABL = 0, int CAP = TileCap<T>::value, int MAXWG = kMergeMaxWgs, bool RA = (OSP_RANK_ATOMIC != 0), bool GA = false>
__global__ __launch_bounds__(NT, (merge_waves_per_simd<T, NT, CAP, MAXWG, ABL>())) void merge_tiles_kernel(
    const TileDesc *__restrict__ desc, uint32_t ntiles, const MergeLevels<T> lvl, int colbits, uint64_t *tile_status,
    uint32_t *ticket, const uint64_t *__restrict__ out_base_p, uint32_t *__restrict__ c_col, T *__restrict__ c_val,
    uint64_t *__restrict__ out_end_p, const ChunkTable<T> ct = ChunkTable<T>{}, uint32_t nshards = 1, uint32_t *abort_word = nullptr,
    const GatherArgs<T> ga = GatherArgs<T>{}) {
    __shared__ MergeSmem<T, NT, CAP> sm;
    __shared__ TileDesc s_dnext;
    __shared__ uint32_t s_tnext;
    // Mode 64 (row-wise variant): level-0 tiles -- rows that fit a tile -- are not read from the staging buffer; their
    // partial products are computed here from the chunk table, and their values stay in this array
    constexpr bool ROWWISE = (ABL & 64) != 0;
    __shared__ T rwval[ROWWISE ? CAP : 1];
    constexpr int kTileCap = CAP;
    constexpr int NW = NT / kWave;
    constexpr int LPT = (kTileCap + NT - 1) / NT;
    constexpr int ITERS = (kTileCap / NW + kWave - 1) / kWave;  // wave iterations per sort pass
    constexpr int IPT = LPT;
    constexpr int DPT = (kDigits + NT - 1) / NT;  // digits per thread in the scan step
    static_assert(kTileMaxRows + 1 <= NT, "row offsets are fetched one per thread");
    unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    uint32_t shard = 0;   // (thread 0's copy is the one in use: it takes every ticket of the workgroup)
    if constexpr (GA) { if (tid < (uint32_t)kTileCap / 32u) sm.gbits[tid] = 0; }   // (the barrier below)
    if (tid == 0) {
        if (nshards > 1) shard = atomicAdd(&ticket[nshards * kTicketStride], 1u) % nshards;   // by arrival, see take_ticket
        const uint32_t t0 = (ABL & 4) ? blockIdx.x : take_ticket(ticket, nshards, shard, ntiles);
        s_tnext = t0;
        if (t0 < ntiles) s_dnext = desc[t0];
    }
    __syncthreads();
    // A workgroup that starts late -- the CUs were still busy with the previous kernel, or with another process -- can
    // find every tile already taken by the workgroups that started on time.  Its descriptor slot was never filled:
    // leave before anything reads it.  (Round 1's intermittent "memory access fault": the garbage level index of
    // that slot indexed the kernel-argument arrays.)
#ifndef OSP_TEST_NO_LATE_GUARD  // (defined only to show that tests/test_gpu_shared_device.py catches the old behaviour)
    if (s_tnext >= ntiles) return;
#endif
    const uint64_t out_base = (ABL & 32) ? 0ull : *out_base_p;
    const uint32_t colmask = colbits < 32 ? ((1u << colbits) - 1u) : 0xffffffffu;
    // the first tile's data is requested here; inside the loop the NEXT tile's data is requested while the
    // current tile's output is being written
    uint32_t t = s_tnext;
    TileDesc d = s_dnext;
    OSP_CRUMB(5, ntiles, d.lvl, d.nr);
    uint64_t ro = 0;
    PartWords<T> lrec[LPT];  // raw: unpacked at staging time, so the loads stay in flight together
    // a thread's q-th entry of a tile is ix0 + q * ixs: tid + q * NT, or -- in the instantiation that knows gathered tiles --
    // w * kSpan + q * 64 + lane: a wave owns a contiguous span of the tile (as in the sort passes), so that the records a wave
    // gathers in one load are consecutive.  (For a staged tile the two are the same to the memory system: 64 consecutive
    // records per wave and load either way.)
    constexpr uint32_t kSpan = (uint32_t)kTileCap / NW;
    static_assert(!GA || (kTileCap % (NW * kWave) == 0 && (int)kSpan == LPT * kWave), "gathered tiles: a wave's span is LPT wave-loads");
    uint32_t ix0 = GA ? w * kSpan + lane : tid;
    constexpr uint32_t ixs = GA ? (uint32_t)kWave : (uint32_t)NT;
    auto request = [&](const TileDesc &dd, bool ok) {
        bool fetch = ok && dd.n <= (uint32_t)kTileCap;
        ro = 0;
#ifndef OSP_TEST_NO_LATE_GUARD
        const uint32_t lv_i = dd.lvl & 1u;  // two levels; never index the argument arrays with anything else
#else
        const uint32_t lv_i = dd.lvl;
#endif
        const Part<T> *__restrict__ stage = lvl.stage[lv_i];
        if (fetch && tid <= dd.nr) ro = lvl.row_off[lv_i][dd.ra + tid];
        if (ROWWISE && dd.lvl == 0) fetch = false;  // nothing was staged for these rows
        if constexpr (GA) {
            if (fetch && dd.rcnt != 0) {   // (tile-uniform) a gathered tile: nothing of it is in the staging buffers; its run number
                                           // `tid` (dst, src) waits where a staged tile's first record would (the rest, if any, is
                                           // loaded when the tile starts)
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const RunDesc<T> *__restrict__ gruns = dd.lvl ? ga.runs : ga.runs0;
                const u32x2 ds2 = *reinterpret_cast<const u32x2 *>(&gruns[tid < dd.rcnt ? dd.rbeg + tid : dd.rbeg]);
                const u32x2 ds3 = *reinterpret_cast<const u32x2 *>(&gruns[tid + NT < dd.rcnt ? dd.rbeg + tid + NT : dd.rbeg]);   // (rows of many short chunks: 250-400 runs per tile)
                lrec[0].w[0] = ds2.x;
                lrec[0].w[1] = ds2.y;
                lrec[1].w[0] = ds3.x;
                lrec[1].w[1] = ds3.y;
                return;
            }
        }
#pragma unroll
        for (int q = 0; q < LPT; q++) {
            const uint32_t i = ix0 + q * ixs;
            // unconditional load from a clamped address (lanes past the end read the descriptor array and
            // ignore it): a branch here makes the compiler wait for every load inside its own block
            const Part<T> *src = (fetch && i < dd.n) ? &stage[dd.s + i] : reinterpret_cast<const Part<T> *>(desc);
            lrec[q] = load_part_words(src);  // (a non-temporal load here costs the value reload its L2 hits: merge +6 %)
        }
    };
    request(d, t < ntiles);
    __syncthreads();  // everybody holds t / d before the slots are refilled
    OSP_PROF_DECL
    while (t < ntiles) {
        // (The compiler hoists every per-thread index expression of the tile loop -- tid * 6 + q, addresses of the thread's
        // counters ... some thirty values -- out of the loop and, short of registers, SPILLS them: 21 of them with the gathered
        // path compiled in, reloaded sixty times per tile.  Hiding the thread index behind an empty asm once per tile makes
        // them the few integer instructions they are: no spills at the same 96 registers.)
        if constexpr (GA) { asm volatile("" : "+v"(tid)); lane = tid & 63u; w = tid >> 6; ix0 = w * kSpan + lane; }
        OSP_DESC_CHECK(t, d);
        OSP_CRUMB(1, ntiles, d.lvl, d.nr);
        const uint64_t ra = d.ra, s = d.s, base = lvl.base[d.lvl];
        const uint32_t nr = d.nr, n = d.n;
        int64_t *__restrict__ c_rowptr = lvl.c_rowptr[d.lvl];
        const bool long_row = n > (uint32_t)kTileCap;
        // The next tile's ticket is taken only AFTER this tile's look-back has returned: a workgroup that
        // is still waiting for its predecessors must not sit on a ticket, or every later tile queues up
        // behind it (measured: 4x slower merge with an early ticket).
        uint32_t tn_reg = ntiles;
        if (long_row) {
            // a single long row, already reduced by the split / global-sort path: it only takes part in the
            // offset chain; heavy_copy_kernel moves its entries once c_rowptr is known
            if (w == 0) {
                const uint64_t total = lvl.heavy_nnz[d.lvl][ra];
                const uint64_t excl = (ABL & 2) ? (uint64_t)t * kTileCap : lookback_prefix(tile_status, t, total, abort_word);
                if (lane == 0) {
                    tn_reg = (ABL & 4) ? t + gridDim.x : take_ticket(ticket, nshards, shard, ntiles);
                    c_rowptr[ra] = (int64_t)(out_base + excl);
                    if (t + 1 == ntiles) *out_end_p = out_base + excl + total;
                    s_tnext = tn_reg;
                    if (tn_reg < ntiles) s_dnext = desc[tn_reg];
                }
            }
            __syncthreads();
            t = s_tnext;
            d = s_dnext;
            request(d, t < ntiles);
            __syncthreads();
            continue;
        }
        int rowbits = 0;
        while ((1u << rowbits) < nr) rowbits++;
        const bool relkey = d.kbits != 0;  // level-1 tile: key = col - cbase
        const uint32_t cbase = d.cbase;
        const int keybits = relkey ? (int)d.kbits : colbits + rowbits;
        const int nbits = (n && !(ABL & 1)) ? keybits : 0;
        // The tile's entry count (= distinct keys) is needed by every later tile's look-back.  Waiting for
        // the sort to deliver it makes successors stall behind slower predecessors (measured: +27 %); a hash set
        // over the keys gives the same number right after staging.  The table lives in LDS that is idle until the
        // first sort pass (everything behind key0: exactly 2.5 words per entry).
        constexpr uint32_t HS = MergeSmem<T, NT, CAP>::kHashWords;
        typedef MergeSmem<T, NT, CAP> Smem;
        static_assert(!OSP_HASH_IN_CNT || offsetof(Smem, cnt) + sizeof(sm.cnt) == offsetof(Smem, pos0) + 4u * HS,
                      "hash table: from pos0 through the digit counters without a gap");
        uint32_t *htab = sm.htab();
        constexpr bool INPLACE = (ABL & 32) != 0;
        const bool early = nbits > 0 && keybits < 32 && !(ABL & 2) && !(ABL & 8) && !INPLACE;
        const bool gath = GA && d.rcnt != 0;   // (tile-uniform)
        uint32_t kq[LPT];
        T vq[LPT];   // the values by staging position: read back after the sort -- or, in a gathered tile, formed below and kept
        // gathered tile: the run of each of the thread's entries (16 bits each) -- through the hash count in registers, through the
        // sort in `pad` (idle until the values arrive; f32 tiles have none and keep the registers)
        uint32_t rq[(LPT + 1) / 2];
        constexpr bool kRunInPad = GA && Smem::kWide;
        // ... and (f64 tiles: the kernel has the registers) the entries' positions in B, so that after the sort B's value and
        // the run's A value are fetched side by side instead of one after the other
        constexpr bool kKeepPos = GA && Smem::kWide;
        uint32_t bpos[LPT];
        // ---- a gathered tile: its records are formed here (cscMulcsr for these rows) ----
        // Its runs go to LDS first: position in B minus the run's start inside the tile (gsrc: where the hash table's first words
        // would be -- the table of such a tile is what lies behind), and one BIT per run start (sm.gbits: all clear between
        // tiles).  Entry i belongs to the last run that starts at or before i = (run starts up to i) - 1: a wave owns a
        // contiguous span of the tile, so that is a running count plus a population count inside the 64 bits of the wave's
        // current window -- no search.  The entry's column is fetched now, its value after the sort (the sort needs the keys
        // only, and a thread cannot hold six values through it).
        constexpr uint32_t GW = GA ? (uint32_t)kTileCap : 0u;   // words of gsrc
        constexpr uint32_t HSG = HS - GW;                       // words of a gathered tile's hash table
        uint32_t *gsrc = htab, *htab_g = htab + GW;
        if constexpr (GA) {
            if (gath) {
                static_assert(HS > 2 * GW, "gathered tiles: the run table leaves a hash table of more than two words per entry... of half the entries");
                const uint32_t R = d.rcnt, tile0 = (uint32_t)(s + base);   // (the buffers' positions are 32 bits wide, modulo)
                const RunDesc<T> *__restrict__ gruns = d.lvl ? ga.runs : ga.runs0;
                for (uint32_t x = tid; x < R; x += NT) {
                    uint32_t dst = lrec[0].w[0], src = lrec[0].w[1];
                    if (x == tid + NT) { dst = lrec[1].w[0]; src = lrec[1].w[1]; }
                    else if (x != tid) { const RunDesc<T> rd = gruns[d.rbeg + x]; dst = rd.dst; src = rd.src; }
                    const uint32_t st = dst - tile0;
                    gsrc[x] = src - st;
                    atomicOr(&sm.gbits[st >> 5], 1u << (st & 31u));
                }
                if (early) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    static_assert((offsetof(Smem, pos0) + 4u * GW) % 16 == 0, "gathered tiles: the hash table starts 16-byte aligned");
                    constexpr uint32_t nvec = HSG / 4, tail = HSG % 4;
                    u32x4 *hv = reinterpret_cast<u32x4 *>(htab_g);
                    const u32x4 ones = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                    for (uint32_t i = tid; i < nvec; i += NT) hv[i] = ones;
                    if (tid < tail) htab_g[nvec * 4 + tid] = 0xffffffffu;
                    if (tid == 0) sm.hcount = 0;
                }
            }
        }
        if (!gath && early) {
            if constexpr (ABL & 256) {
                for (uint32_t i = tid; i < HS; i += NT) htab[i] = 0xffffffffu;
            } else {
                // 16 bytes per lane and store: a quarter of the LDS store instructions (the table starts 8-byte aligned
                // behind key0, so the first and last words are written singly)
                static_assert((kTileCap * 4) % 8 == 0, "hash table: 16-byte groups behind an 8-byte aligned start (odd words at both ends singly)");
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                constexpr uint32_t lead = ((kTileCap * 4) % 16) / 4;          // words up to the first 16-byte boundary
                constexpr uint32_t nvec = (HS - lead) / 4, tail = (HS - lead) % 4;
                u32x4 *hv = reinterpret_cast<u32x4 *>(htab + lead);
                const u32x4 ones = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                for (uint32_t i = tid; i < nvec; i += NT) hv[i] = ones;
                if (tid < lead) htab[tid] = 0xffffffffu;
                if (tid < tail) htab[lead + nvec * 4 + tid] = 0xffffffffu;
            }
            if (tid == 0) sm.hcount = 0;
        }
        if (tid <= nr) sm.rowo[tid] = (uint32_t)(ro - base - s);
        __syncthreads();
        OSP_PROF_MARK(0);
        // stage the keys: (local row << colbits) | col, or col - cbase.  The payload (staging position) is implicit
        // until pass 0; the values stay in registers until the sort is over.
        uint32_t fresh = 0;
        const bool rw_tile = ROWWISE && d.lvl == 0;
        if constexpr (GA) {
            if (gath) {
                // run starts before the wave's span (wave-uniform), then window by window
                uint32_t cnt0 = 0;
                {
                    const uint32_t nwords = w * (kSpan / 32u);   // <= 3 * 14 words: one per lane
                    static_assert((NW - 1) * (kSpan / 32u) <= (uint32_t)kWave, "gathered tiles: the bits before a wave's span, one word per lane");
                    cnt0 = wave_reduce_sum<uint32_t>(lane < nwords ? (uint32_t)__popc(sm.gbits[lane]) : 0u);
                    cnt0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt0);
                }
#pragma unroll
                for (int q = 0; q < (LPT + 1) / 2; q++) rq[q] = 0;
#pragma unroll
                for (int q = 0; q < LPT; q++) {
                    const uint32_t i = ix0 + q * ixs;
                    const uint32_t wi = w * (kSpan / 32u) + 2u * (uint32_t)q;   // the window's two words (wave-uniform address)
                    const uint32_t mlo = sm.gbits[wi], mhi = sm.gbits[wi + 1];
                    // starts at or before my position inside the window: those below my lane, and mine
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                    const uint32_t own = ((lane < 32 ? mlo >> lane : mhi >> (lane - 32)) & 1u);
                    const uint32_t r = cnt0 + below + own - 1u;
                    cnt0 += (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
                    bpos[q] = i < n ? gsrc[r] + i : 0u;   // clamped: the loads below go out together
                    rq[q >> 1] |= (i < n ? r : 0u) << (16 * (q & 1));
                }
                uint32_t bc[LPT];
#pragma unroll
                for (int q = 0; q < LPT; q++) bc[q] = ga.b_colidx[bpos[q]];
#pragma unroll
                for (int q = 0; q < LPT; q++) {
                    const uint32_t i = ix0 + q * ixs;
                    if (relkey) {
                        kq[q] = bc[q] - cbase;
                    } else {   // short rows: the entry's row inside the tile is the key's major part
                        uint32_t lo = 0, hi = nr;  // last r with rowo[r] <= i
                        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sm.rowo[mid] <= i) lo = mid; else hi = mid; }
                        kq[q] = (colbits < 32 ? (lo << colbits) : 0u) | bc[q];
                    }
                    if (i < n) sm.key0[i] = kq[q];
                }
                OSP_PROF_MARK(11);   // gathered tile: run lookup, columns from B
                if (early) {
                    uint32_t hq[LPT], oq[LPT];
#pragma unroll
                    for (int q = 0; q < LPT; q++) {
                        hq[q] = (uint32_t)(((uint64_t)(kq[q] * 2654435761u) * HSG) >> 32);
                        oq[q] = kq[q];
                        if (ix0 + q * ixs < n) oq[q] = atomicCAS(&htab_g[hq[q]], 0xffffffffu, kq[q]);
                    }
#pragma unroll
                    for (int q = 0; q < LPT; q++) {
                        if (ix0 + q * ixs < n) {
                            uint32_t old = oq[q], h = hq[q];
                            while (old != 0xffffffffu && old != kq[q]) {
                                h = (h + 1 == HSG) ? 0u : h + 1;
                                old = atomicCAS(&htab_g[h], 0xffffffffu, kq[q]);
                            }
                            fresh += old == 0xffffffffu;
                        }
                    }
                }
            }
        }
        if (gath) {
            // (formed above)
        } else if (rw_tile) {
            if constexpr (ROWWISE) {
                // Row-wise tile: the partial products of rows [ra, ra+nr) are formed here.  The rows' chunks (one per
                // non-zero A[i,k], in (row, k) order -- the staging order) are taken kRwGroup at a time by a wave: lane q
                // holds chunk q's start position, B row and A value; then the lanes walk the group's entries, each
                // finding its chunk by bisecting the start positions with ds_bpermute.  Four steps of the walk are in
                // flight at once (their gathers from B are independent).
                // (16 chunks per group keeps all four waves busy on tiles of ~100 chunks; a wave-full of 64 for tiles of tiny
                // chunks was measured and made no difference)
                constexpr uint32_t kRwGroup = 16, kRwSteps = 4, kRwUnroll = 4;
                const uint32_t c0 = ct.rowfirst[ra], c1 = ct.rowfirst[ra + nr];
                for (uint32_t cb = c0 + w * kRwGroup; cb < c1; cb += NW * kRwGroup) {
                    const uint32_t c = cb + lane;
                    const bool cv = lane < kRwGroup && c < c1;
                    const uint64_t o0 = cv ? ct.off[c] : 0ull, o1 = cv ? ct.off[c + 1] : 0ull;
                    const uint32_t st = cv ? (uint32_t)(o0 - base - s) : n;  // lanes past the end: behind every entry
                    const uint32_t en = cv ? (uint32_t)(o1 - base - s) : n;
                    const uint32_t bsv = cv ? ct.bs[c] : 0u;
                    const T av = cv ? ct.a_vals[ct.perm[c]] : T(0);
                    uint32_t lrow = 0;
                    {
                        uint32_t lo = 0, hi = nr;  // last r with rowo[r] <= st
                        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sm.rowo[mid] <= st) lo = mid; else hi = mid; }
                        lrow = colbits < 32 ? (lo << colbits) : 0u;
                    }
                    const uint32_t nvalid = min(kRwGroup, c1 - cb);
                    const uint32_t gbeg = wave_bcast(st, 0u), gend = wave_bcast(en, nvalid - 1);
                    for (uint32_t ib0 = gbeg; ib0 < gend; ib0 += kRwUnroll * kWave) {
                        uint32_t bcol[kRwUnroll], crow[kRwUnroll];
                        T bval[kRwUnroll], cav[kRwUnroll];
#pragma unroll
                        for (uint32_t u = 0; u < kRwUnroll; u++) {
                            const uint32_t i = ib0 + u * kWave + lane;
                            uint32_t lo = 0, hi = kRwGroup;  // last lane whose chunk starts at or before i
#pragma unroll
                            for (uint32_t step = 0; step < kRwSteps; step++) {
                                const uint32_t mid = (lo + hi) >> 1;
                                const uint32_t sv = (uint32_t)__shfl((int)st, (int)mid);
                                if (sv <= i) lo = mid; else hi = mid;
                            }
                            const uint32_t cst = (uint32_t)__shfl((int)st, (int)lo);
                            const uint32_t cbs = (uint32_t)__shfl((int)bsv, (int)lo);
                            crow[u] = (uint32_t)__shfl((int)lrow, (int)lo);
                            cav[u] = __shfl(av, (int)lo);
                            const uint32_t b = i < gend ? cbs + (i - cst) : 0u;  // clamped, branch-free: loads stay in flight
                            bcol[u] = ct.b_colidx[b];
                            bval[u] = ct.b_vals[b];
                        }
#pragma unroll
                        for (uint32_t u = 0; u < kRwUnroll; u++) {
                            const uint32_t i = ib0 + u * kWave + lane;
                            if (i < gend) {
                                const uint32_t key = crow[u] | bcol[u];
                                sm.key0[i] = key;
                                rwval[i] = cav[u] * bval[u];
                                if (early) {
                                    uint32_t h = (uint32_t)(((uint64_t)(key * 2654435761u) * HS) >> 32);
                                    uint32_t old = atomicCAS(&htab[h], 0xffffffffu, key);
                                    while (old != 0xffffffffu && old != key) {
                                        h = (h + 1 == HS) ? 0u : h + 1;
                                        old = atomicCAS(&htab[h], 0xffffffffu, key);
                                    }
                                    fresh += old == 0xffffffffu;
                                }
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < LPT; q++) {
                const uint32_t i = ix0 + q * ixs;
                kq[q] = 0;
                if (i < n) {
                    if (relkey) {
                        kq[q] = lrec[q].col() - cbase;
                    } else {
                        uint32_t lo = 0, hi = nr;  // last r with rowo[r] <= i
                        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sm.rowo[mid] <= i) lo = mid; else hi = mid; }
                        kq[q] = (colbits < 32 ? (lo << colbits) : 0u) | lrec[q].col();
                    }
                    sm.key0[i] = kq[q];
                }
            }
            if (early) {
                // first probes of all the thread's keys go out together; only collisions with a different key walk on
                uint32_t hq[LPT], oq[LPT];
#pragma unroll
                for (int q = 0; q < LPT; q++) {
                    hq[q] = (uint32_t)(((uint64_t)(kq[q] * 2654435761u) * HS) >> 32);
                    oq[q] = kq[q];
                    if (ix0 + q * ixs < n) oq[q] = atomicCAS(&htab[hq[q]], 0xffffffffu, kq[q]);
                }
#pragma unroll
                for (int q = 0; q < LPT; q++) {
                    if (ix0 + q * ixs < n) {
                        uint32_t old = oq[q], h = hq[q];
                        while (old != 0xffffffffu && old != kq[q]) {
                            h = (h + 1 == HS) ? 0u : h + 1;
                            old = atomicCAS(&htab[h], 0xffffffffu, kq[q]);
                        }
                        fresh += old == 0xffffffffu;
                    }
                }
            }
        }
        if (early) {
            const uint32_t wsum = wave_incl_scan(fresh);  // DPP: no LDS round trips; lane 63 holds the wave's total
            if (lane == kWave - 1 && wsum) atomicAdd(&sm.hcount, wsum);
        }
        // each wave ranks a contiguous span, so earlier waves = earlier positions (stable)
        const uint32_t per = (n + NW - 1) / NW;
        const uint32_t wbeg = min(w * per, n), wend = min(wbeg + per, n);
        int cur = 0;
        __syncthreads();   // keys staged, hash count complete
        if constexpr (GA) {
            if (gath) {
                if (tid < (uint32_t)kTileCap / 32u) sm.gbits[tid] = 0;   // (every thread has read them: the barrier above)
                if constexpr (kRunInPad) {
#pragma unroll
                    for (int q = 0; q < LPT; q++) {
                        const uint32_t i = ix0 + q * ixs;
                        if (i < n) sm.pad[i] = (uint16_t)(rq[q >> 1] >> (16 * (q & 1)));
                    }
                }
            }
        }
        OSP_PROF_MARK(1);
        if (early && tid == 0) lookback_publish(tile_status, t, sm.hcount);
        OSP_PROF_MARK(2);
        const int npass = (nbits + kDigitBits - 1) / kDigitBits;
        const int pbits = npass ? (nbits + npass - 1) / npass : 0;  // balanced digit width (<= kDigitBits)
        const uint32_t dmask = (1u << pbits) - 1u;
        const int ndig = 1 << pbits;                                // digits in use: counters beyond are not touched
        for (int pass = 0, shift = 0; pass < npass; pass++, shift += pbits) {
            uint32_t *ksrc = sm.key(cur), *kdst = sm.key(cur ^ 1);
            uint16_t *psrc = sm.pos(cur), *pdst = sm.pos(cur ^ 1);
            if constexpr (ABL & 128) {
                for (int dd = lane; dd < ndig; dd += kWave) sm.cnt[w][dd] = 0;
            } else {
                // four 16-bit counters per lane and store (the rows are 8-byte aligned; a short digit range zeroes a few
                // counters beyond it, which nobody reads)
                uint64_t *c64 = reinterpret_cast<uint64_t *>(sm.cnt[w]);
                for (int dd = lane; dd < (ndig + 3) / 4; dd += kWave) c64[dd] = 0ull;
            }
            // (a) rank inside the wave's span; keys and ranks stay in registers for (c)
            uint32_t kreg[ITERS], rreg[ITERS];
            auto rank_span = [&](auto bits_tag) {
                constexpr int BITS = decltype(bits_tag)::value;
#pragma unroll
                for (int it = 0; it < ITERS; it++) {
                    const uint32_t i = wbeg + it * kWave + lane;
                    const bool valid = i < wend;
                    kreg[it] = valid ? ksrc[i] : 0u;
                    const unsigned dg = (kreg[it] >> shift) & dmask;
                    unsigned rk, cntd;
                    wave_match_digit<BITS>(dg, valid, rk, cntd);
                    rreg[it] = 0;
                    if (valid) {
                        const uint32_t c = sm.cnt[w][dg];
                        rreg[it] = c + rk;
                        if (rk == 0) sm.cnt[w][dg] = (uint16_t)(c + cntd);
                    }
                }
            };
            if constexpr (RA) {
            // Rank by LDS atomic: one ds_add_rtn on the wave's packed 16-bit counter returns "entries of this digit so far",
            // and lanes that hit the same counter in one instruction get their old values in ascending lane order -- which
            // is exactly the stable rank.  (The order is not documented; tools/test_lds_atomic_order checked 1.8e10 ranks on
            // gfx950 against the ballot match below, and every parity test compares values bit for bit, which an unstable
            // rank would break.)
            {
                uint32_t(*cnt32)[kDigits / 2] = reinterpret_cast<uint32_t(*)[kDigits / 2]>(sm.cnt);
#pragma unroll
                for (int it = 0; it < ITERS; it++) {
                    const uint32_t i = wbeg + it * kWave + lane;
                    const bool valid = i < wend;
                    kreg[it] = valid ? ksrc[i] : 0u;
                    const unsigned dg = (kreg[it] >> shift) & dmask;
                    const unsigned half = 16u * (dg & 1u);
                    rreg[it] = 0;
                    if (valid) rreg[it] = (atomicAdd(&cnt32[w][dg >> 1], 1u << half) >> half) & 0xffffu;
                }
                (void)rank_span;
            }
            } else {
            // the widest digits only when they save a pass (20 key bits in two passes); else one ballot less
            if (pbits > kDigitBits - 1) rank_span(std::integral_constant<int, kDigitBits>{});
            else rank_span(std::integral_constant<int, kDigitBits - 1>{});
            }
            __syncthreads();
            OSP_PROF_MARK(3);
            // (b) exclusive scan over (digit major, wave minor); a thread owns dpt consecutive digits
            if constexpr ((ABL & 128) != 0 || DPT != 4) {
                const int dpt = (ndig + NT - 1) / NT;  // (bounds-checked below: NT need not be a power of two)
                uint32_t c[DPT][NW], ssum = 0;
#pragma unroll
                for (int q = 0; q < DPT; q++) {
                    const int dg = tid * dpt + q;
                    const bool on = q < dpt && dg < ndig;
#pragma unroll
                    for (int ww = 0; ww < NW; ww++) { c[q][ww] = on ? sm.cnt[ww][dg] : 0u; ssum += c[q][ww]; }
                }
                uint32_t total;
                uint32_t ex = block_excl_scan<uint32_t, NT, false>(ssum, sm.scratch, &total)  /* the barrier after the counter update follows */;
#pragma unroll
                for (int q = 0; q < DPT; q++) {
                    const int dg = tid * dpt + q;
                    if (q < dpt && dg < ndig) {
#pragma unroll
                        for (int ww = 0; ww < NW; ww++) { sm.cnt[ww][dg] = (uint16_t)ex; ex += c[q][ww]; }
                    }
                }
            } else {
                // A thread owns FOUR consecutive digits whatever the digit range (threads beyond it idle): the four 16-bit
                // counters of one wave are one aligned 8-byte word -- one LDS load and one LDS store per wave row instead of
                // four each.
                const bool on = (int)tid * 4 < ndig;
                uint64_t pk[NW];
                uint32_t ssum = 0;
#pragma unroll
                for (int ww = 0; ww < NW; ww++) {
                    pk[ww] = on ? reinterpret_cast<const uint64_t *>(sm.cnt[ww])[tid] : 0ull;
                    // sum of the four 16-bit fields (each < 2^16, the sum < 2^18)
                    ssum += (uint32_t)(pk[ww] & 0xffffu) + (uint32_t)((pk[ww] >> 16) & 0xffffu) + (uint32_t)((pk[ww] >> 32) & 0xffffu) +
                            (uint32_t)(pk[ww] >> 48);
                }
                uint32_t total;
                uint32_t ex = block_excl_scan<uint32_t, NT, false>(ssum, sm.scratch, &total)  /* the barrier after the counter update follows */;
                if (on) {
                    // digit-major, wave-minor exclusive offsets
                    uint32_t o[4][NW];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
#pragma unroll
                        for (int ww = 0; ww < NW; ww++) { o[q][ww] = ex; ex += (uint32_t)(pk[ww] >> (16 * q)) & 0xffffu; }
                    }
#pragma unroll
                    for (int ww = 0; ww < NW; ww++)
                        reinterpret_cast<uint64_t *>(sm.cnt[ww])[tid] = (uint64_t)(o[0][ww] & 0xffffu) | ((uint64_t)(o[1][ww] & 0xffffu) << 16) |
                                                                        ((uint64_t)(o[2][ww] & 0xffffu) << 32) | ((uint64_t)(o[3][ww] & 0xffffu) << 48);
                }
            }
            __syncthreads();
            OSP_PROF_MARK(4);
            // (c) scatter
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const uint32_t i = wbeg + it * kWave + lane;
                if (i < wend) {
                    const uint32_t k = kreg[it];
                    const uint32_t dst = (uint32_t)sm.cnt[w][(k >> shift) & dmask] + rreg[it];
                    kdst[dst] = k;
                    pdst[dst] = pass == 0 ? (uint16_t)i : psrc[i];
                }
            }
            cur ^= 1;
            __syncthreads();
            OSP_PROF_MARK(5);
        }
        uint32_t *skey = sm.key(cur);   // sorted keys and their staging positions
        uint16_t *spos = sm.pos(cur);
        T *sval = rw_tile ? rwval : sm.vals(cur);  // the 8 bytes per entry beside them: values by staging position
        if (npass == 0) {  // nothing was sorted (empty tile): the payload is still implicit
            for (uint32_t i = tid; i < n; i += NT) spos[i] = (uint16_t)i;
        }
        // the values: read again (the tile went through L2 a few microseconds ago) rather than held in 18 registers
        // through the sort -- that is what lets five workgroups per CU run without spilling.  The barriers inside the
        // scan below order the LDS writes before the run sums.
        OSP_CRUMB(2, npass, cur, nbits);
        if (!rw_tile && !gath) {
            const Part<T> *__restrict__ stg = lvl.stage[d.lvl];
#pragma unroll
            for (int q = 0; q < LPT; q++) {
                const uint32_t i = ix0 + q * ixs;
                vq[q] = load_part_words(i < n ? &stg[s + i] : reinterpret_cast<const Part<T> *>(desc)).val();  // clamped, branch-free
            }
        }
        if constexpr (GA) {
            if (gath) {
                // the values of a gathered tile: the entry's run (its descriptor comes from L2) -> its place in B -> A value x B value,
                // the product cscMulcsr forms (one rounding, as in multiply_kernel)
                const uint32_t tile0 = (uint32_t)(s + base);
                const RunDesc<T> *__restrict__ gruns = d.lvl ? ga.runs : ga.runs0;
                T bv[LPT];
                if constexpr (kKeepPos) {
                    T av[LPT];
#pragma unroll
                    for (int q = 0; q < LPT; q++) {
                        const uint32_t i = ix0 + q * ixs;
                        bv[q] = ga.b_vals[bpos[q]];
                        av[q] = gruns[d.rbeg + (i < n ? (uint32_t)sm.pad[i] : 0u)].av;
                    }
#pragma unroll
                    for (int q = 0; q < LPT; q++) vq[q] = av[q] * bv[q];
                    (void)tile0;
                } else {
                RunDesc<T> rdq[LPT];
#pragma unroll
                for (int q = 0; q < LPT; q++) {
                    const uint32_t i = ix0 + q * ixs;
                    uint32_t r;
                    if constexpr (kRunInPad) r = i < n ? (uint32_t)sm.pad[i] : 0u;
                    else r = (rq[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                    rdq[q] = gruns[d.rbeg + r];
                }
#pragma unroll
                for (int q = 0; q < LPT; q++) {
                    const uint32_t i = ix0 + q * ixs;
                    bv[q] = ga.b_vals[i < n ? rdq[q].src - (rdq[q].dst - tile0) + i : 0u];   // clamped, branch-free
                }
#pragma unroll
                for (int q = 0; q < LPT; q++) vq[q] = rdq[q].av * bv[q];
                }
            }
        }
        if (npass == 0) __syncthreads();
        // head flags + exclusive scan (blocked: thread owns IPT consecutive sorted entries) -- while the values are on
        // their way
        const uint32_t ib = tid * IPT;
        uint32_t heads = 0, hmask = 0;
#pragma unroll
        for (int q = 0; q < IPT; q++) {
            const uint32_t i = ib + q;
            if (i < n) {
                const bool h = (i == 0) || (skey[i] != skey[i - 1]);
                hmask |= (h ? 1u : 0u) << q;
                heads += h;
            }
        }
        if constexpr (kRunInPad) {
            if (gath) __syncthreads();   // every thread has read its runs from `pad`: the values may take its place
        }
        if (!rw_tile) {
#pragma unroll
            for (int q = 0; q < LPT; q++) {
                const uint32_t i = ix0 + q * ixs;
                if (i < n) sval[i] = vq[q];
            }
        }
        if (tid == 0) sm.nlongrun = 0;  // (the scan's barrier orders this before the run sums)
        uint32_t total;
        uint32_t ex = block_excl_scan<uint32_t, NT, false>(heads, sm.scratch, &total);  // (the next scan is barriers away)
        OSP_PROF_MARK(6);
        // the tile's unique count is known: wave 0 runs the look-back and then requests the next ticket;
        // the ticket's round trip overlaps the run sums below
        if (w == 0) {
            const uint64_t excl = INPLACE   ? 0ull
                                 : (ABL & 2) ? (uint64_t)t * kTileCap
                                 : early     ? lookback_prefix<false>(tile_status, t, total, abort_word)  // count already published
                                             : lookback_prefix<true>(tile_status, t, total, abort_word);
            if (lane == 0) {
                sm.excl = excl;
                tn_reg = (ABL & 4) ? t + gridDim.x : take_ticket(ticket, nshards, shard, ntiles);
            }
        }
        OSP_PROF_MARK(7);
        const uint32_t oslot0 = ex;   // output slot of the first run that starts among the thread's entries
#pragma unroll
        for (int q = 0; q < IPT; q++) {
            const uint32_t i = ib + q;
            if (i < n) {
                sm.rank[i] = (uint16_t)ex;  // output slot of the run that starts at/behind i (the counters are dead)
                ex += (hmask >> q) & 1u;
            }
        }
        if (tid == 0) sm.rank[n] = (uint16_t)total;
        // each head sums its run in staging order (= ascending k)
        // A head walks at most kRunShort entries itself.  A longer run (a hub column: at Graph500 skew one output entry of a
        // tile is fed by hundreds of products) would keep ONE lane busy for three dependent LDS round trips per entry while
        // the rest of the workgroup waits; it is handed, with the sum so far, to a whole wave below.
        // A thread's IPT sorted entries are in its registers first (independent gathers, all in flight together); the runs
        // that lie inside them are summed there, left to right, with the head flags telling where a run ends -- no key
        // compares, no loop of dependent LDS reads per run.  Only the LAST run of the block can reach into the entries of the
        // next threads: its sum goes on from LDS, up to kRunShort entries from its head; what is longer than that goes to a
        // whole wave below.  (Runs were walked entry by entry from LDS before: tiles of products that compress 2:1 spent a
        // quarter of their time there, tools/bench_merge DUPWIN=230.)  Results sit at the run's last entry of the block.
        static_assert(IPT <= kRunShort, "the part of a run inside one thread's entries is never longer than a head's own share");
        T acc[IPT];
        uint32_t ocol[IPT], lmask = 0, tmask = 0;
        {
            T v[IPT];
#pragma unroll
            for (int q = 0; q < IPT; q++) {
                const uint32_t i = ib + q;
                v[q] = i < n ? sval[spos[i]] : T(0);
            }
            T cur = T(0);
#pragma unroll
            for (int q = 0; q < IPT; q++) {
                const uint32_t i = ib + q;
                const bool hd = (hmask >> q) & 1u;
                cur = hd ? v[q] : cur + v[q];   // (the sum STARTS as the head's value: keeps a lone -0.0)
                const bool nexthd = q + 1 < IPT ? ((hmask >> (q + 1)) & 1u) != 0 : true;
                const bool tail = i < n && (i + 1 >= n || nexthd) && (hmask & ((2u << q) - 1u)) != 0;   // of a run that started in this block
                acc[q] = cur;
                ocol[q] = 0;
                tmask |= (tail ? 1u : 0u) << q;
            }
        }
        if (tmask) {
            const int qt = 31 - __builtin_clz(tmask);                 // the block's last run ends (so far) here
            const uint32_t it = ib + (uint32_t)qt;
            if (qt == IPT - 1 && it + 1 < n) {                        // it may go on behind the block
                const uint32_t ih = ib + (uint32_t)(31 - __builtin_clz(hmask));   // its head
                const uint32_t k = skey[it];
                T a = acc[IPT - 1];
                const uint32_t ulim = min(n, ih + (uint32_t)kRunShort);
                uint32_t u = it + 1;
                for (; u < ulim && skey[u] == k; u++) a += sval[spos[u]];
                if (u == ulim && u < n && skey[u] == k) {
                    const uint32_t idx = atomicAdd(&sm.nlongrun, 1u);
                    sm.long_sum()[idx] = a;
                    sm.long_pos()[idx] = (uint16_t)u;
                    lmask |= 1u << (IPT - 1);
                    a = index_as_value<T>(idx);
                }
                acc[IPT - 1] = a;
            }
#pragma unroll
            for (int q = 0; q < IPT; q++) {
                if ((tmask >> q) & 1u) {
                    const uint32_t k = skey[ib + q];
                    ocol[q] = relkey ? k + cbase : (k & colmask);
                }
            }
        }
        if (tid == 0) s_tnext = tn_reg;
        __syncthreads();  // all gathers from the keys / values done; sm.excl and s_tnext are published
        if (const uint32_t nlr = sm.nlongrun) {
            // the long runs, one per wave at a time: 64 values per LDS round trip, then added one by one IN ORDER (the sum
            // is the same left-to-right chain of additions, only fed from registers)
            T *lsum = sm.long_sum();
            const uint16_t *lpos = sm.long_pos();
            for (uint32_t r = w; r < nlr; r += NW) {
                const uint32_t u1 = lpos[r], k = skey[u1];
                T a = lsum[r];
                for (uint32_t u0 = u1;; u0 += kWave) {
                    const uint32_t u = u0 + lane;
                    const bool in = u < n && skey[u] == k;
                    const T v = in ? sval[spos[u]] : T(0);
                    const uint64_t m = __ballot(in);
                    const uint32_t c = m == ~0ull ? (uint32_t)kWave : (uint32_t)__builtin_ctzll(~m);  // the run is contiguous
                    for (uint32_t x = 0; x < c; x++) a += wave_bcast(v, x);
                    if (c < (uint32_t)kWave) break;
                }
                if (lane == 0) lsum[r] = a;
            }
            __syncthreads();  // before the compaction overwrites what the waves are still reading
        }
        OSP_PROF_MARK(8);
        // compact in LDS (columns over the sorted keys, sums over the values), then stream out with consecutive
        // lanes on consecutive addresses
#pragma unroll
        for (int q = 0; q < IPT; q++) {
            if ((tmask >> q) & 1u) {   // the run that ends at the thread's entry q: its slot is its head's
                const uint32_t slot = oslot0 + (uint32_t)__popc(hmask & ((2u << q) - 1u)) - 1u;
                skey[slot] = ocol[q];
                sval[slot] = ((lmask >> q) & 1u) ? sm.long_sum()[value_as_index<T>(acc[q])] : acc[q];
            }
        }
        if (tid == 0 && tn_reg < ntiles) s_dnext = desc[tn_reg];
        __syncthreads();
        OSP_PROF_MARK(9);
        const uint32_t tn = s_tnext;
        const TileDesc dn = s_dnext;
        OSP_CRUMB(3, total, tn, dn.s);
        // the next tile's HBM reads go out ahead of this tile's writes
        request(dn, tn < ntiles);
        if constexpr (INPLACE) {
            // the segment's merged entries go back over its own beginning (everything of it is in LDS by now)
            Part<T> *outp = const_cast<Part<T> *>(lvl.stage[d.lvl]) + s;
            for (uint32_t o = tid; o < total; o += NT) outp[o] = Part<T>{skey[o], sval[o]};
            if (tid == 0) const_cast<uint32_t *>(lvl.heavy_nnz[d.lvl])[ra] = total;
        } else {
            const uint64_t obase = out_base + sm.excl;
            for (uint32_t o = tid; o < total; o += NT) {  // the result is not read again by this product: streaming stores
                __builtin_nontemporal_store(skey[o], &c_col[obase + o]);
                __builtin_nontemporal_store(sval[o], &c_val[obase + o]);
            }
            // rows keep their index span through the sort (row is the major key)
            if (tid < nr) c_rowptr[ra + tid] = (int64_t)(obase + sm.rank[sm.rowo[tid]]);
            if (t + 1 == ntiles && tid == 0) *out_end_p = obase + total;
        }
        OSP_CRUMB(4, total, tn, 0);
        t = tn;
        d = dn;
        __syncthreads();  // LDS is reused by the next tile
        OSP_PROF_MARK(10);
    }
    OSP_CRUMB(7, ntiles, 0, 0);
    OSP_PROF_FLUSH;
}

// ---- merge: over-long segments -------------------------------------------------------------------
// Segments (column ranges of split long rows) that still exceed a tile.  Those up to kBigTileCap entries -- in
// R-MAT products nearly all of them: the lowest column range of a mid-sized row, where the hub columns sit -- are
// merged by the SAME tile kernel instantiated with a larger tile (in-place mode, 512 threads); the rest (one output
// entry fed by many thousands of products) take the global-sort path below.
constexpr int kBigTileCap = 4096;
constexpr int kBigTileThreads = 512;
// list[t] -> huge[scan[t]] or mid[t - scan[t]] by length; scan = exclusive scan of the "huge" flags
struct SegHugeFlag {
    const uint32_t *list;
    const uint64_t *row_off;
    uint32_t cap;
    __device__ uint32_t operator()(uint64_t t) const {
        const uint32_t v = list[t];
        return (row_off[v + 1] - row_off[v]) > (uint64_t)cap ? 1u : 0u;
    }
};
__global__ void seg_partition_kernel(SegHugeFlag f, const uint32_t *scan, uint32_t n, uint32_t *huge, uint32_t *mid) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (f(t)) huge[scan[t]] = f.list[t]; else mid[t - scan[t]] = f.list[t];
}
// one in-place tile per mid-sized over-long segment v: key = col - first column of the segment
__global__ void seg_tile_desc_kernel(const uint32_t *mid, uint32_t nmid, const uint64_t *vrow_off, const uint32_t *vcol0,
                                     const uint32_t *vcol1, TileDesc *desc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nmid) return;
    const uint32_t v = mid[t];
    TileDesc d;
    d.s = vrow_off[v];
    d.ra = v;
    d.nr = 1;
    d.n = (uint32_t)(vrow_off[v + 1] - vrow_off[v]);
    d.lvl = 1;
    d.cbase = vcol0[v];
    const uint64_t range = (uint64_t)(vcol1[v] - vcol0[v]);
    int kb = 1;
    while (kb < 32 && (1ull << kb) < range) kb++;
    d.kbits = (uint32_t)kb;
    d.rbeg = 0;
    d.rcnt = 0;
    desc[t] = d;
}

// ---- merge: global-sort path for segments longer than kBigTileCap --------------------------------
struct HeavyLen {
    const uint32_t *rows;
    const uint64_t *row_off;
    __device__ uint64_t operator()(uint64_t h) const { return row_off[rows[h] + 1] - row_off[rows[h]]; }
};
// key = (heavy rank << colbits) | col, payload = panel-relative staging position
__global__ void heavy_fill_kernel(const uint32_t *rows, const uint64_t *hoff, uint32_t nheavy,
                                  const uint64_t *row_off, uint64_t base, int colbits,
                                  const char *stage, uint32_t rec_bytes, uint64_t nh, uint64_t *key, uint32_t *pos) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nh) return;
    uint64_t h = upper_bound_dev(hoff, 0, (uint64_t)nheavy + 1, x) - 1;
    uint64_t p = row_off[rows[h]] - base + (x - hoff[h]);
    key[x] = (h << colbits) | (uint64_t)(*(const uint32_t *)(stage + p * rec_bytes));
    pos[x] = (uint32_t)p;
}
struct HeavyHeadFlag {
    const uint64_t *key;
    __device__ uint32_t operator()(uint64_t x) const { return (x == 0 || key[x] != key[x - 1]) ? 1u : 0u; }
};
template <class T>
__global__ void heavy_gather_kernel(const uint32_t *pos, const Part<T> *stage, uint64_t nh, T *sorted_val) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < nh) sorted_val[x] = stage[pos[x]].val;
}
// head_pos[r] = first sorted position of run r (r = headscan at a head); head_pos[number of runs] = nh
__global__ void heavy_heads_kernel(const uint64_t *key, const uint64_t *headscan, uint64_t nh, uint64_t *head_pos) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x > nh) return;
    if (x == nh) { head_pos[headscan[nh]] = nh; return; }
    if (x == 0 || key[x] != key[x - 1]) head_pos[headscan[x]] = x;
}
// One thread per run of equal (segment, col) keys: the run is summed in sorted order (= ascending staging position
// = ascending k) and written in place.  Short runs are summed by their own thread; a run of more than 64 entries --
// a pile: one output entry fed by thousands of products -- is summed by the whole wave: 64 values per coalesced load,
// added one after the other in the same order (v_readlane), so the result does not change, only the memory access.
template <class T>
__global__ __launch_bounds__(256) void heavy_reduce_kernel(const uint64_t *key, const T *sorted_val, const uint64_t *headscan,
                                                           const uint64_t *head_pos, uint64_t nh, const uint32_t *rows,
                                                           const uint64_t *hoff, uint32_t nheavy, const uint64_t *row_off,
                                                           uint64_t base, int colbits, Part<T> *stage) {
    constexpr uint64_t kWaveRun = 64;
    const uint64_t nruns = headscan[nh];
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = lane_id();
    const bool mine = r < nruns;
    uint64_t a = 0, e = 0;
    if (mine) { a = head_pos[r]; e = head_pos[r + 1]; }
    T acc = 0;
    const bool lng = mine && (e - a) > kWaveRun;
    if (mine && !lng) {
        acc = sorted_val[a];
        for (uint64_t u = a + 1; u < e; u++) acc += sorted_val[u];
    }
    uint64_t m = __ballot(lng);
    while (m) {
        const uint32_t j = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const uint64_t ja = wave_bcast(a, j), je = wave_bcast(e, j);
        T sum = 0;
        for (uint64_t b0 = ja; b0 < je; b0 += kWave) {
            const uint64_t x = b0 + lane;
            const T v = x < je ? sorted_val[x] : (T)0;
            const uint32_t cnt = (uint32_t)min((uint64_t)kWave, je - b0);
            uint32_t t0 = 0;
            if (b0 == ja) { sum = wave_bcast(v, 0u); t0 = 1; }  // the sum starts AS the first entry (keeps a lone -0.0)
            for (uint32_t t = t0; t < cnt; t++) sum += wave_bcast(v, t);
        }
        if (lane == j) acc = sum;
    }
    if (!mine) return;
    const uint64_t k = key[a];
    const uint64_t h = k >> colbits;
    const uint64_t o = row_off[rows[h]] - base + (r - headscan[hoff[h]]);
    stage[o] = Part<T>{(uint32_t)(k & ((1ull << colbits) - 1ull)), acc};
}
__global__ void heavy_rows_kernel(const uint32_t *rows, const uint64_t *hoff, uint32_t nheavy,
                                  const uint64_t *headscan, uint32_t *heavy_nnz) {
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nheavy) return;
    heavy_nnz[rows[h]] = (uint32_t)(headscan[hoff[h + 1]] - headscan[hoff[h]]);
}
// after merge_tiles_kernel fixed c_rowptr: move each reduced segment's merged entries to their place.
// Two launches.  Most of these segments are dense accumulators' results or big in-place tiles (at most a few thousand
// entries) and there are hundreds of thousands of them per panel at Graph500 skew: eight workgroups per segment (as it was until
// late in round 3) were mostly workgroups with nothing to do -- 4.4 ms per panel went into dispatching them.  Now a WAVE copies
// the first kHeavyCopyHead entries of a segment (two segments per wave, eight per workgroup), and a second launch, one
// workgroup per segment, copies what lies beyond (the results of the global-sort path) and returns at once otherwise.
constexpr uint32_t kHeavyCopyHead = 8192;
template <class T>
__global__ __launch_bounds__(256) void heavy_copy_kernel(const uint32_t *rows, uint32_t nheavy, const uint64_t *heavy_src,
                                                         const uint32_t *heavy_nnz, const int64_t *c_rowptr, const Part<T> *src_stage,
                                                         const uint32_t *scol, const T *sval, uint32_t *c_col, T *c_val) {
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
#pragma unroll
    for (uint32_t j = 0; j < 2; j++) {
        const uint32_t h = (blockIdx.x * 4u + w) * 2u + j;
        if (h >= nheavy) return;
        const uint32_t row = rows[h];
        const uint64_t n = min((uint64_t)heavy_nnz[row], (uint64_t)kHeavyCopyHead), src = heavy_src[h], dst = (uint64_t)c_rowptr[row];
        for (uint64_t i = lane; i < n; i += kWave) {
            if (src_stage) { const Part<T> pp = src_stage[src + i]; c_col[dst + i] = pp.col; c_val[dst + i] = pp.val; }
            else { c_col[dst + i] = scol[src + i]; c_val[dst + i] = sval[src + i]; }
        }
    }
}
template <class T>
__global__ __launch_bounds__(256) void heavy_copy_rest_kernel(const uint32_t *rows, uint32_t nheavy, const uint64_t *heavy_src,
                                                              const uint32_t *heavy_nnz, const int64_t *c_rowptr, const Part<T> *src_stage,
                                                              const uint32_t *scol, const T *sval, uint32_t *c_col, T *c_val) {
    const uint32_t h = blockIdx.x;
    if (h >= nheavy) return;
    const uint32_t row = rows[h];
    const uint64_t n = heavy_nnz[row];
    if (n <= kHeavyCopyHead) return;
    const uint64_t src = heavy_src[h], dst = (uint64_t)c_rowptr[row];
    for (uint64_t i = kHeavyCopyHead + threadIdx.x; i < n; i += blockDim.x) {
        if (src_stage) { const Part<T> pp = src_stage[src + i]; c_col[dst + i] = pp.col; c_val[dst + i] = pp.val; }
        else { c_col[dst + i] = scol[src + i]; c_val[dst + i] = sval[src + i]; }
    }
}
__global__ void heavy_src_inplace_kernel(const uint32_t *rows, uint32_t nheavy, const uint64_t *row_off, uint64_t base,
                                         uint64_t *heavy_src) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < nheavy) heavy_src[h] = row_off[rows[h]] - base;
}

// ---- output bound -------------------------------------------------------------------------------
// nnz(C) <= sum_i min(U_i, N): capacity of the final CSR arrays
struct RowUpperBound {
    const uint64_t *row_off;  // already offset to the first row of the range
    uint64_t N;
    __device__ uint64_t operator()(uint64_t r) const { return min(row_off[r + 1] - row_off[r], N); }
};
// Estimated cost of row r, in sixteenths of "one partial product of a short row": a row that fits a tile costs
// multiply + merge; a long row also pays the column-range split (x1.6), a row beyond the one-workgroup split the
// two-pass stretch split (x2.5, which also absorbs what else is slower about hub rows).  (Fitted to the per-shard times of tools/shard_balance.py on R-MAT-22 mild: 18,
// 29 and 35 ps per partial product.)  Only the balance of the row shards depends on it, never a result.
struct RowCost {
    const uint64_t *row_off;
    uint64_t cap, one_wg_max;
    __device__ uint64_t operator()(uint64_t r) const {
        const uint64_t u = row_off[r + 1] - row_off[r];
        return u * (u <= cap ? 16u : u <= one_wg_max ? 26u : 40u);
    }
};
// bounds[j] = first row whose cost prefix reaches j/G of the total (j = 0..G): row ranges of equal estimated work
// that every rank computes identically from the operands alone; offs[j] = staging offset of that row
__global__ void shard_bounds_kernel(const uint64_t *cost_pre, const uint64_t *row_off, uint64_t M, uint32_t G, uint64_t *bounds,
                                    uint64_t *offs) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > G) return;
    const uint64_t total = cost_pre[M];
    uint64_t r = j == G ? M : lower_bound_dev(cost_pre, 0, M + 1, (uint64_t)((unsigned __int128)total * j / G));
    if (r > M) r = M;
    bounds[j] = r;
    offs[j] = row_off[r];
}
__global__ void set_u64_kernel(uint64_t *p, uint64_t v) { *p = v; }

// ---- row-sharded mode: this rank's rows only, before the symbolic phase ---------------------------------------
// work[r] += stride * nnz(B[k,:]) for the non-zeros A[r,k] of every stride-th column of the k shard: an estimate of
// each row's partial products from a deterministic sample of the columns (integer atomics: order-free, so every rank
// gets the same numbers).  One wave per sampled column.  Only the BALANCE of the row shards depends on the estimate.
__global__ void row_work_kernel(const int64_t *a_colptr, const uint32_t *a_rowidx, const int64_t *b_rowptr, uint64_t k0,
                                uint64_t k1, uint32_t stride, unsigned long long *work) {
    const uint64_t wv = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t k = k0 + wv * stride;
    if (k >= k1) return;
    const unsigned long long w = (unsigned long long)(b_rowptr[k + 1] - b_rowptr[k]) * stride;
    if (!w) return;
    const int64_t lo = a_colptr[k], hi = a_colptr[k + 1];
    for (int64_t e = lo + lane_id(); e < hi; e += kWave) atomicAdd(&work[a_rowidx[e]], w);
}
// A restricted to rows [r0, r1) by stream compaction of its non-zeros (CSC order is kept, so it stays a CSC):
// keep[t] flags, their exclusive scan = new positions; the new column pointers are the scan read at the old ones
struct RowInRange {
    const uint32_t *rows;  // already offset to the k shard's first non-zero
    uint32_t r0;
    uint64_t r1;
    __device__ uint32_t operator()(uint64_t t) const { const uint32_t r = rows[t]; return (r >= r0 && (uint64_t)r < r1) ? 1u : 0u; }
};
__global__ void restrict_colptr_kernel(const int64_t *a_colptr, uint64_t K, int64_t e0, uint64_t nnz, const uint32_t *keep_scan,
                                       int64_t *colptr2) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > K) return;
    int64_t t = a_colptr[k] - e0;
    t = t < 0 ? 0 : (t > (int64_t)nnz ? (int64_t)nnz : t);  // columns outside the k shard become empty
    colptr2[k] = (int64_t)keep_scan[t];
}
template <class T>
__global__ void restrict_compact_kernel(RowInRange keep, const uint32_t *keep_scan, uint64_t nnz, const T *a_vals, uint32_t *rowidx2,
                                        T *vals2) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz || !keep(t)) return;
    const uint32_t j = keep_scan[t];
    rowidx2[j] = keep.rows[t];
    vals2[j] = a_vals[t];
}
struct LoadU64AsI64 {
    const uint64_t *p;
    __device__ int64_t operator()(uint64_t i) const { return (int64_t)p[i]; }
};

// ---- CSR parts -> staging (multi-GPU final merge) ----------------------------------
// Row r of part p becomes one chunk of row r; chunks ordered by p.
struct PartsChunkLen {  // candidate chunk c = r * nparts + p
    const int64_t *const *rowptrs;
    int nparts;
    __device__ uint64_t operator()(uint64_t c) const {
        const uint64_t r = c / (uint64_t)nparts;
        const int p = (int)(c - r * (uint64_t)nparts);
        return (uint64_t)(rowptrs[p][r + 1] - rowptrs[p][r]);
    }
};
__global__ void parts_rows_kernel(const uint64_t *offs, int nparts, uint64_t M, uint64_t *row_off) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > M) return;
    row_off[r] = offs[r * (uint64_t)nparts];
}
template <class T>
__global__ void parts_scatter_kernel(const int64_t *const *rowptrs, const uint32_t *const *colidxs,
                                     const T *const *valss, int nparts, uint64_t r0, uint64_t r1,
                                     const uint64_t *row_off, uint64_t base, Part<T> *stage) {
    // one wave per row
    const uint64_t r = r0 + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    if (r >= r1) return;
    uint64_t dst = row_off[r] - base;
    for (int p = 0; p < nparts; p++) {
        const int64_t b = rowptrs[p][r], e = rowptrs[p][r + 1];
        for (int64_t i = b + lane_id(); i < e; i += kWave) {
            stage[dst + (i - b)] = Part<T>{colidxs[p][i], valss[p][i]};
        }
        dst += (uint64_t)(e - b);
    }
}

// the same for parts given as packed records (osp_merge_record_parts): a part's row is copied as it stands
template <class T>
__global__ void parts_scatter_rec_kernel(const int64_t *const *rowptrs, const Part<T> *const *recs, int nparts, uint64_t r0, uint64_t r1,
                                         const uint64_t *row_off, uint64_t base, Part<T> *stage) {
    const uint64_t r = r0 + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    if (r >= r1) return;
    const unsigned lane = lane_id();
    uint64_t dst = row_off[r] - base;
    // the row's extent in up to 64 parts at a time is fetched by one lane per part (one round trip instead of a chain of
    // dependent loads per part), then every part's piece is copied with four loads per lane in flight
    for (int p0 = 0; p0 < nparts; p0 += kWave) {
        const int np = min(nparts - p0, (int)kWave);
        int64_t b_l = 0, e_l = 0;
        if ((int)lane < np) { b_l = rowptrs[p0 + lane][r]; e_l = rowptrs[p0 + lane][r + 1]; }
        for (int q = 0; q < np; q++) {
            const uint64_t b = wave_bcast((uint64_t)b_l, (uint32_t)q), len = wave_bcast((uint64_t)(e_l - b_l), (uint32_t)q);
            const Part<T> *src = recs[p0 + q] + b;
            Part<T> *out = stage + dst;
            constexpr int U = 4;
            for (uint64_t i0 = 0; i0 < len; i0 += (uint64_t)U * kWave) {
                PartWords<T> w[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint64_t i = i0 + (uint64_t)u * kWave + lane;
                    w[u] = load_part_words(&src[i < len ? i : 0]);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint64_t i = i0 + (uint64_t)u * kWave + lane;
                    if (i < len) store_part_words(&out[i], w[u]);
                }
            }
            dst += len;
        }
    }
}

}  // namespace osp
