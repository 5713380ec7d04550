// osp_prims.h -- device-wide building blocks for gfx950 (wave64): exclusive scan and a stable
// LSD radix sort of (key, u32 payload) pairs.  Hand-written; no rocPRIM / hipCUB.
//
// These serve the symbolic phase (CSC -> row-ordered chunk offsets) and the global-sort merge
// path.  The reference has no counterpart: its coo2csr (SimSpGEMM.cpp:102-152) and
// deduplicateCOO (:519-535) call std::sort on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Stable radix ranks by LDS atomic (1) or by ballot matching (0).  One ds_add_rtn instruction hands its old values to
// the lanes that hit the same counter in ascending lane order -- undocumented, so tools/test_lds_atomic_order checks it
// (1.8e10 ranks on gfx950, none out of order), every parity test compares values bit for bit, and every context tests
// it when it is created: BOTH variants of every ranking kernel are compiled (template parameter RA), and a context whose
// self-test fails runs the ballot ones.  The macro only sets what a context prefers (0 = ballot always: `make ballot`).
#ifndef OSP_RANK_ATOMIC
#define OSP_RANK_ATOMIC 1
#endif

namespace osp {

constexpr int kWave = 64;
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

template <class T>
__device__ __forceinline__ T wave_incl_scan(T v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T o = __shfl_up(v, d, kWave);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}
// 32-bit inclusive scan on the DPP path (no LDS crossbar): Hillis-Steele inside each row of 16
// lanes (row_shr 1,2,4,8 with zero fill), then row_bcast:15 / row_bcast:31 to carry across rows.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_fetch(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
// the value of the lane below (lane 0: 0) on the DPP path: wave_shr:1.  (__shfl_up is a ds_bpermute: an LDS operation, and
// the wait for it also waits for every LDS atomic still in flight.)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return dpp_fetch<0x138, 0xf>(v); }
template <>
__device__ __forceinline__ uint32_t wave_incl_scan<uint32_t>(uint32_t v) {
    v += dpp_fetch<0x111, 0xf>(v);  // row_shr:1
    v += dpp_fetch<0x112, 0xf>(v);  // row_shr:2
    v += dpp_fetch<0x114, 0xf>(v);  // row_shr:4
    v += dpp_fetch<0x118, 0xf>(v);  // row_shr:8
    v += dpp_fetch<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v += dpp_fetch<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}

template <class T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_down(v, d, kWave);
    return __shfl(v, 0, kWave);
}

// Sum of a 64-bit value over the wave, for every lane, without the LDS crossbar: three DPP scans over 24-bit slices
// (each slice's sum over 64 lanes fits 32 bits).  Values must be below 2^62.
__device__ __forceinline__ uint64_t wave_reduce_sum_u62(uint64_t v) {
    const uint32_t a = wave_incl_scan((uint32_t)(v & 0xffffffu));
    const uint32_t b = wave_incl_scan((uint32_t)((v >> 24) & 0xffffffu));
    const uint32_t c = wave_incl_scan((uint32_t)(v >> 48));
    const uint64_t sa = (uint32_t)__builtin_amdgcn_readlane((int)a, 63), sb = (uint32_t)__builtin_amdgcn_readlane((int)b, 63),
                   sc = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
    return sa + (sb << 24) + (sc << 48);
}

// Exclusive scan across a block of NT threads (NT multiple of 64, <= 1024).  `scratch` holds
// NT/64 entries of T.  Returns the exclusive prefix of `v`; *total gets the block sum.
// TAILSYNC = false leaves out the barrier that protects `scratch` against the caller's NEXT scan: for callers that pass
// another barrier of their own before they scan again.
template <class T, int NT, bool TAILSYNC = true>
__device__ __forceinline__ T block_excl_scan(T v, T *scratch, T *total) {
    constexpr int NW = NT / kWave;
    const T incl = wave_incl_scan(v);
    const int w = threadIdx.x >> 6;
    if (lane_id() == kWave - 1) scratch[w] = incl;
    __syncthreads();
    T off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const T tv = scratch[i];
        if (i < w) off += tv;
        tot += tv;
    }
    *total = tot;
    if (TAILSYNC) __syncthreads();  // scratch may be reused by the caller's next scan
    return incl - v + off;
}

// ---------------------------------------------------------------------------------------
// Device-wide exclusive scan: out[i] = sum_{j<i} f(j) for i in [0, n]; out has n+1 entries.
// Three launches: tile sums, scan of tile sums (one block), tile scan + offset.
// ---------------------------------------------------------------------------------------
template <class F, class TOut>
__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(F f, uint64_t n,
                                                                      TOut *tile_sums) {
    __shared__ TOut scratch[kScanThreads / kWave + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    TOut s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        uint64_t idx = base + (uint64_t)i * kScanThreads + threadIdx.x;
        if (idx < n) s += (TOut)f(idx);
    }
    TOut total;
    block_excl_scan<TOut, kScanThreads>(s, scratch, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

template <class TOut>
__global__ __launch_bounds__(kScanThreads) void scan_tile_offsets_kernel(TOut *tile_sums,
                                                                         uint64_t ntiles) {
    __shared__ TOut scratch[kScanThreads / kWave + 1];
    TOut carry = 0;
    for (uint64_t b = 0; b < ntiles; b += kScanThreads) {
        uint64_t i = b + threadIdx.x;
        TOut v = i < ntiles ? tile_sums[i] : (TOut)0;
        TOut total;
        TOut ex = block_excl_scan<TOut, kScanThreads>(v, scratch, &total);
        if (i < ntiles) tile_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

template <class F, class TOut>
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(F f, uint64_t n,
                                                                  const TOut *tile_offs,
                                                                  TOut *out) {
    __shared__ TOut scratch[kScanThreads / kWave + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    // blocked arrangement: thread t owns items [t*kScanItems, (t+1)*kScanItems)
    TOut v[kScanItems];
    TOut s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * kScanItems + i;
        v[i] = idx < n ? (TOut)f(idx) : (TOut)0;
        s += v[i];
    }
    TOut total;
    TOut ex = block_excl_scan<TOut, kScanThreads>(s, scratch, &total) + tile_offs[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * kScanItems + i;
        if (idx < n) out[idx] = ex;
        ex += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = tile_offs[gridDim.x];
}

template <class TOut>
__global__ void scan_empty_kernel(TOut *out) { out[0] = 0; }

// Short inputs (the per-long-row and per-coarse-block arrays of a product: a few hundred to a few thousand entries): one
// workgroup walks the tiles with a running carry -- one launch instead of three.  In-place use (out == the array f reads)
// is fine: a tile's inputs are in registers before its outputs are written.
constexpr uint64_t kScanSmallTiles = 16;
template <class F, class TOut>
__global__ __launch_bounds__(kScanThreads) void scan_small_kernel(F f, uint64_t n, TOut *out) {
    __shared__ TOut scratch[kScanThreads / kWave + 1];
    TOut carry = 0;
    for (uint64_t base = 0; base < n; base += kScanTile) {
        TOut v[kScanItems];
        TOut s = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; i++) {
            const uint64_t idx = base + (uint64_t)threadIdx.x * kScanItems + i;
            v[i] = idx < n ? (TOut)f(idx) : (TOut)0;
            s += v[i];
        }
        TOut total;
        TOut ex = block_excl_scan<TOut, kScanThreads>(s, scratch, &total) + carry;
#pragma unroll
        for (int i = 0; i < kScanItems; i++) {
            const uint64_t idx = base + (uint64_t)threadIdx.x * kScanItems + i;
            if (idx < n) out[idx] = ex;
            ex += v[i];
        }
        carry += total;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// `tile_scratch` needs ceil(n / kScanTile) + 1 entries of TOut.
template <class F, class TOut>
inline void device_exclusive_scan(F f, uint64_t n, TOut *out, TOut *tile_scratch,
                                  hipStream_t stream) {
    if (n == 0) {
        scan_empty_kernel<TOut><<<1, 1, 0, stream>>>(out);
        return;
    }
    const uint64_t ntiles = (n + kScanTile - 1) / kScanTile;
    if (ntiles <= kScanSmallTiles) {
        scan_small_kernel<F, TOut><<<1, kScanThreads, 0, stream>>>(f, n, out);
        return;
    }
    scan_tile_sums_kernel<F, TOut><<<(unsigned)ntiles, kScanThreads, 0, stream>>>(f, n, tile_scratch);
    scan_tile_offsets_kernel<TOut><<<1, kScanThreads, 0, stream>>>(tile_scratch, ntiles);
    scan_tiles_kernel<F, TOut><<<(unsigned)ntiles, kScanThreads, 0, stream>>>(f, n, tile_scratch, out);
}
inline uint64_t scan_scratch_entries(uint64_t n) { return (n + kScanTile - 1) / kScanTile + 2; }

// ---------------------------------------------------------------------------------------
// Stable LSD radix sort of (key, u32 payload) pairs, 8-bit digits.
//   per pass: histogram (digit-major [256][nblocks]) -> exclusive scan -> stable scatter.
// Ranking inside a block is by wave-level match (8 ballots) so equal digits keep input order.
// ---------------------------------------------------------------------------------------
constexpr int kSortThreads = 256;
constexpr int kSortRounds = 16;
constexpr int kSortTile = kSortThreads * kSortRounds;  // 4096 elements per block
constexpr int kRadix = 256;

// Lanes of this wave holding the same 8-bit digit (only among `valid` lanes).
__device__ __forceinline__ uint64_t wave_match8(unsigned digit, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

template <class K>
__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const K *keys, uint64_t n,
                                                                 int shift, uint32_t *hist,
                                                                 uint32_t nblocks) {
    __shared__ uint32_t h[kRadix];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
#pragma unroll 4
    for (int r = 0; r < kSortRounds; r++) {
        uint64_t i = base + (uint64_t)r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <class K, bool RA>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(
    const K *keys_in, const uint32_t *vals_in, K *keys_out, uint32_t *vals_out, uint64_t n,
    int shift, const uint32_t *hist_scan, uint32_t nblocks) {
    constexpr int NW = kSortThreads / kWave;
    __shared__ uint32_t base[kRadix];     // next free global slot per digit for this block
    __shared__ uint32_t cnt[NW][kRadix];  // per-round, per-wave digit counts
    base[threadIdx.x] = hist_scan[(uint64_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < NW; w++) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * kSortTile;
    for (int r = 0; r < kSortRounds; r++) {
        const uint64_t i = tile + (uint64_t)r * kSortThreads + threadIdx.x;
        const bool valid = i < n;
        K key = 0;
        uint32_t val = 0;
        if (valid) {
            key = keys_in[i];
            val = vals_in[i];
        }
        const unsigned d = (unsigned)(key >> shift) & 255u;
        unsigned rank = 0;
        if constexpr (RA) {  // stable rank by LDS atomic: old values come back in lane order (see merge_tiles_kernel)
            if (valid) rank = atomicAdd(&cnt[w][d], 1u);
        } else {
            const uint64_t peers = wave_match8(d, valid);
            rank = __popcll(peers & lanemask_lt());
            if (valid && rank == 0) cnt[w][d] = (uint32_t)__popcll(peers);
        }
        __syncthreads();
        if (valid) {
            uint32_t off = base[d] + rank;
#pragma unroll
            for (int ww = 0; ww < NW; ww++)
                if (ww < w) off += cnt[ww][d];
            keys_out[off] = key;
            vals_out[off] = val;
        }
        __syncthreads();
        {
            uint32_t s = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) {
                s += cnt[ww][threadIdx.x];
                cnt[ww][threadIdx.x] = 0;
            }
            base[threadIdx.x] += s;
        }
        __syncthreads();
    }
}

struct LoadU32 {
    const uint32_t *p;
    __device__ uint32_t operator()(uint64_t i) const { return p[i]; }
};

inline uint64_t sort_blocks(uint64_t n) { return (n + kSortTile - 1) / kSortTile; }
// u32 entries needed for the histogram (+1) and for its scan scratch.
inline uint64_t sort_hist_entries(uint64_t n) { return sort_blocks(n) * kRadix + 1; }

// Sorts bits [0, nbits) of the keys.  keys/vals ping-pong between buffer 0 and 1; returns the
// index (0/1) of the buffer holding the sorted data.  n must be < 2^32.
template <class K>
inline int device_radix_sort_pairs(K *keys[2], uint32_t *vals[2], uint64_t n, int nbits,
                                   uint32_t *hist, uint32_t *scan_scratch, hipStream_t stream, bool rank_atomic) {
    int cur = 0;
    if (n == 0) return cur;
    const uint32_t nblocks = (uint32_t)sort_blocks(n);
    for (int shift = 0; shift < nbits; shift += 8) {
        sort_hist_kernel<K><<<nblocks, kSortThreads, 0, stream>>>(keys[cur], n, shift, hist, nblocks);
        // in-place exclusive scan of the digit-major histogram (reads precede writes per tile)
        device_exclusive_scan<LoadU32, uint32_t>(LoadU32{hist}, (uint64_t)nblocks * kRadix, hist,
                                                 scan_scratch, stream);
        if (rank_atomic)
            sort_scatter_kernel<K, true><<<nblocks, kSortThreads, 0, stream>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift,
                                                                                hist, nblocks);
        else
            sort_scatter_kernel<K, false><<<nblocks, kSortThreads, 0, stream>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift,
                                                                                 hist, nblocks);
        cur ^= 1;
    }
    return cur;
}

}  // namespace osp
