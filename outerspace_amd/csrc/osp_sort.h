// osp_sort.h -- the symbolic phase's sort: stable LSD radix sort of (row, A-entry) pairs, 8-bit
// digits, written for gfx950 (wave64, 160 KiB LDS).
//
// Why it exists: the staging layout needs the non-zeros of A (given in CSC order) in (row, k) order.
// The reference gets its orderings from std::sort on the host (coo2csr, SimSpGEMM.cpp:111-121).
//
// Per pass: one histogram launch (per-workgroup digit counts, digit-major), one device scan, one
// scatter launch.  A workgroup owns 8192 consecutive elements; ranking is per-wave match (8 ballots)
// with wave-private LDS counters; the elements are first reordered inside LDS so that every digit's
// run leaves the workgroup as one contiguous, coalesced store.
//   * the FIRST pass reads its keys straight from A's row indices (payload = position, implicit),
//   * the LAST pass hands every sorted element to an epilogue functor (here: chunk length lookup),
// so no (key, payload) arrays are materialised before the first or after the last pass.
#pragma once
#include "osp_kernels.h"

namespace osp {

constexpr int kRsThreads = 512;
constexpr int kRsItems = 16;
constexpr int kRsTile = kRsThreads * kRsItems;  // 8192 elements per workgroup

inline uint32_t rs_blocks(uint64_t n) { return (uint32_t)((n + kRsTile - 1) / kRsTile); }
inline uint64_t rs_hist_entries(uint64_t n) { return (uint64_t)rs_blocks(n) * kRadix + 1; }

// FIRST: keys_in is the raw key array and the payload is the element index
template <bool FIRST>
__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint32_t *__restrict__ keys_in, uint64_t n, int shift,
                                                             uint32_t *__restrict__ hist, uint32_t nblocks) {
    __shared__ uint32_t h[kRadix];
    if (threadIdx.x < kRadix) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kRsTile;
#pragma unroll
    for (int q = 0; q < kRsItems; q++) {
        const uint64_t i = base + (uint64_t)q * kRsThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys_in[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kRadix) hist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

struct RsNoEpilogue {
    __device__ void operator()(uint64_t, uint32_t, uint32_t) const {}
};

template <bool FIRST, bool LAST, bool RA, class Epi>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, uint64_t n, int shift, const uint32_t *__restrict__ hist_scan, uint32_t nblocks,
    Epi epi) {
    constexpr int NW = kRsThreads / kWave;
    constexpr int SPAN = kRsTile / NW;      // contiguous elements per wave
    constexpr int ITERS = SPAN / kWave;     // 16
    __shared__ uint32_t skey[kRsTile];
    __shared__ uint32_t sval[kRsTile];
    __shared__ alignas(8) uint16_t cnt[NW][kRadix];
    __shared__ uint32_t lstart[kRadix];     // first LDS slot of each digit
    __shared__ uint32_t gbase[kRadix];      // first global slot of each digit for this workgroup
    __shared__ uint32_t scratch[NW + 1];
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * kRsTile;
    for (int d = lane; d < kRadix; d += kWave) cnt[w][d] = 0;
    if (tid < kRadix) gbase[tid] = hist_scan[(uint64_t)tid * nblocks + blockIdx.x];
    // wave w owns elements [tile + w*SPAN, +SPAN): earlier waves = earlier elements (stable)
    uint32_t kreg[ITERS], vreg[ITERS], rreg[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint64_t i = tile + (uint64_t)w * SPAN + (uint64_t)it * kWave + lane;
        kreg[it] = 0; vreg[it] = 0;
        if (i < n) {
            kreg[it] = keys_in[i];
            vreg[it] = FIRST ? (uint32_t)i : vals_in[i];
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint64_t i = tile + (uint64_t)w * SPAN + (uint64_t)it * kWave + lane;
        const bool valid = i < n;
        const unsigned d = (kreg[it] >> shift) & 255u;
        rreg[it] = 0;
        if constexpr (RA) {
            // stable rank by LDS atomic on the wave's packed 16-bit counter (see merge_tiles_kernel)
            const unsigned half = 16u * (d & 1u);
            if (valid) rreg[it] = (atomicAdd(&reinterpret_cast<uint32_t *>(cnt[w])[d >> 1], 1u << half) >> half) & 0xffffu;
        } else {
            const uint64_t peers = wave_match8(d, valid);
            const unsigned rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            if (valid) {
                const uint32_t c = cnt[w][d];
                rreg[it] = c + rk;
                if (rk == 0) cnt[w][d] = (uint16_t)(c + (uint32_t)__popcll(peers));
            }
        }
    }
    __syncthreads();
    {   // digit totals -> LDS start of every digit, and per-wave offsets inside the digit
        uint32_t c[NW], ssum = 0;
        if (tid < kRadix) {
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { c[ww] = cnt[ww][tid]; ssum += c[ww]; }
        }
        uint32_t total;
        uint32_t ex = block_excl_scan<uint32_t, kRsThreads>(ssum, scratch, &total);
        if (tid < kRadix) {
            lstart[tid] = ex;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { cnt[ww][tid] = (uint16_t)ex; ex += c[ww]; }
        }
    }
    __syncthreads();
    // reorder inside LDS
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint64_t i = tile + (uint64_t)w * SPAN + (uint64_t)it * kWave + lane;
        if (i < n) {
            const uint32_t lp = (uint32_t)cnt[w][(kreg[it] >> shift) & 255u] + rreg[it];
            skey[lp] = kreg[it];
            sval[lp] = vreg[it];
        }
    }
    __syncthreads();
    // stream out: consecutive threads -> consecutive LDS slots -> consecutive global slots per digit
    const uint32_t cnt_here = (uint32_t)min((uint64_t)kRsTile, n - tile);
#pragma unroll
    for (int q = 0; q < kRsItems; q++) {
        const uint32_t j = tid + q * kRsThreads;
        if (j < cnt_here) {
            const uint32_t k = skey[j], v = sval[j];
            const unsigned d = (k >> shift) & 255u;
            const uint64_t g = (uint64_t)gbase[d] + (j - lstart[d]);
            if (LAST) {
                epi(g, k, v);
            } else {
                keys_out[g] = k;
                vals_out[g] = v;
            }
        }
    }
}

// Sort (keys_raw[i], i) by the low `nbits` bits of the key.  Buffers a and b are (key, payload) pairs
// of n entries each; the last pass calls epi(sorted position, key, payload) instead of storing.
// hist needs rs_hist_entries(n) u32, scan_scratch scan_scratch_entries(rs_hist_entries(n)).
template <bool RA, class Epi>
inline void device_sort_rows_ra(const uint32_t *keys_raw, uint64_t n, int nbits, uint32_t *ka, uint32_t *pa, uint32_t *kb,
                                uint32_t *pb, uint32_t *hist, uint32_t *scan_scratch, Epi epi, hipStream_t stream,
                                const uint32_t *vals_raw) {
    if (n == 0) return;
    const uint32_t nb = rs_blocks(n);
    const int npass = std::max(1, (nbits + 7) / 8);
    const uint32_t *kin = keys_raw, *pin = vals_raw;
    uint32_t *kout = ka, *pout = pa;
    for (int p = 0; p < npass; p++) {
        const bool first = p == 0 && vals_raw == nullptr, last = p == npass - 1;
        const int shift = 8 * p;
        if (first) rs_hist_kernel<true><<<nb, kRsThreads, 0, stream>>>(kin, n, shift, hist, nb);
        else rs_hist_kernel<false><<<nb, kRsThreads, 0, stream>>>(kin, n, shift, hist, nb);
        device_exclusive_scan<LoadU32, uint32_t>(LoadU32{hist}, (uint64_t)nb * kRadix, hist, scan_scratch, stream);
        if (first && last) rs_scatter_kernel<true, true, RA, Epi><<<nb, kRsThreads, 0, stream>>>(kin, pin, kout, pout, n, shift, hist, nb, epi);
        else if (first) rs_scatter_kernel<true, false, RA, Epi><<<nb, kRsThreads, 0, stream>>>(kin, pin, kout, pout, n, shift, hist, nb, epi);
        else if (last) rs_scatter_kernel<false, true, RA, Epi><<<nb, kRsThreads, 0, stream>>>(kin, pin, kout, pout, n, shift, hist, nb, epi);
        else rs_scatter_kernel<false, false, RA, Epi><<<nb, kRsThreads, 0, stream>>>(kin, pin, kout, pout, n, shift, hist, nb, epi);
        kin = kout; pin = pout;
        if (kout == ka) { kout = kb; pout = pb; } else { kout = ka; pout = pa; }
    }
}

// rank_atomic: which instantiation of the stable rank runs (Context::rank_atomic)
template <class Epi>
inline void device_sort_rows(const uint32_t *keys_raw, uint64_t n, int nbits, uint32_t *ka, uint32_t *pa, uint32_t *kb,
                             uint32_t *pb, uint32_t *hist, uint32_t *scan_scratch, Epi epi, hipStream_t stream, bool rank_atomic,
                             const uint32_t *vals_raw = nullptr /* payload of keys_raw[i]; default: i itself */) {
    if (rank_atomic) device_sort_rows_ra<true, Epi>(keys_raw, n, nbits, ka, pa, kb, pb, hist, scan_scratch, epi, stream, vals_raw);
    else device_sort_rows_ra<false, Epi>(keys_raw, n, nbits, ka, pa, kb, pb, hist, scan_scratch, epi, stream, vals_raw);
}

// plain epilogue: store the sorted pairs
struct RsStoreEpilogue {
    uint32_t *keys_out, *vals_out;
    __device__ void operator()(uint64_t t, uint32_t k, uint32_t v) const { keys_out[t] = k; vals_out[t] = v; }
};

// ---- on-device COO -> CSR/CSC (coo2csr<transpose> + dupcheck, SimSpGEMM.cpp:102-152, 43-53) ----------------
// out[t] = in[perm[t]]
__global__ void ingest_gather_u32_kernel(const uint32_t *in, const uint32_t *perm, uint64_t n, uint32_t *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[perm[t]];
}
// final gather in (segment, inner) order + range check + adjacent-duplicate check (the reference's dupcheck)
template <class T>
__global__ void ingest_finish_kernel(const uint32_t *seg_sorted, const uint32_t *perm, const uint32_t *inner, const T *vals,
                                     uint64_t n, uint64_t nseg, uint64_t ninner, uint32_t *idx, T *out_vals, uint32_t *flags) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t sgm = seg_sorted[t], in = inner[perm[t]];
    idx[t] = in;
    out_vals[t] = vals[perm[t]];
    if (sgm >= nseg || in >= ninner) atomicOr(flags, kFlagRange);
    if (t > 0 && seg_sorted[t - 1] == sgm && inner[perm[t - 1]] == in) atomicOr(flags, kFlagDuplicate);
}
__global__ void ingest_ptr_kernel(const uint32_t *seg_sorted, uint64_t n, uint64_t nseg, int64_t *ptr) {
    const uint64_t sg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sg <= nseg) ptr[sg] = (int64_t)lower_bound_dev(seg_sorted, 0, n, sg);
}

// ---- symbolic epilogue: for the t-th A entry in (row, k) order ------------------------------------
// rows_sorted[t] = its row, perm[t] = its CSC position (relative to e0), w_sorted[t] = its chunk length.
// (The chunk lengths are looked up in CSC order beforehand -- sym_chunk_len_kernel -- where neighbouring
// threads share a column; doing the column search here, in row order, cost 7.5 ms instead of 0.4.)
struct SymEpilogue {
    const uint32_t *w;   // chunk length per A entry, CSC order -- or, with the B rows wanted, (length, first entry of the B row)
    const uint32_t *bs;  // non-null: `w` holds pairs (one 8-byte gather per chunk instead of two 4-byte ones from two arrays)
    uint32_t *rows_sorted, *perm, *w_sorted, *bs_sorted;
    // gathered rows (osp_kernels.h): `w` holds quads (length, B row, the entry's A value in vwords = 1 or 2 words) and the value
    // goes along to av_sorted -- the ONE gather by `pos` of this pass brings it, where a later gather of a_vals[perm[t]] by
    // every consumer fetched a line per value (short_runs_kernel: 1.5 ms of the headline)
    void *av_sorted = nullptr;
    uint32_t vwords = 0;
    __device__ void operator()(uint64_t t, uint32_t row, uint32_t pos) const {
        rows_sorted[t] = row;
        perm[t] = pos;
        if (av_sorted) {
            const uint4 q = reinterpret_cast<const uint4 *>(w)[pos];
            w_sorted[t] = q.x;
            bs_sorted[t] = q.y;
            if (vwords == 2) reinterpret_cast<uint2 *>(av_sorted)[t] = make_uint2(q.z, q.w);
            else reinterpret_cast<uint32_t *>(av_sorted)[t] = q.z;
        } else if (bs_sorted) {   // (row-wise variant and direct rows: the chunk table)
            const uint2 p = reinterpret_cast<const uint2 *>(w)[pos];
            w_sorted[t] = p.x;
            bs_sorted[t] = p.y;
        } else {
            w_sorted[t] = w[pos];
        }
    }
};
// w[t] = nnz(B[k,:]) for the t-th non-zero of A's shard (CSC order, column k); with bs != nullptr: w holds the pairs
// (nnz(B[k,:]), b_rowptr[k]) -- 2 * nnz words -- and bs itself is only the switch
// a_vals_raw != nullptr: quads (nnz(B[k,:]), b_rowptr[k], the entry's value in vwords words) -- 4 * nnz words
__global__ void sym_chunk_len_kernel(const int64_t *a_colptr, const int64_t *b_rowptr, uint64_t k0, uint64_t k1, int64_t e0,
                                     uint64_t nnz, uint32_t *w, uint32_t *bs, const uint32_t *a_vals_raw = nullptr, uint32_t vwords = 0) {
    // the block's 256 consecutive entries lie in a short range of columns: two full searches per block (first and last
    // entry, a wave each), then every thread bisects that range only (~4 steps instead of ~22)
    __shared__ uint64_t krange[2];
    const uint64_t tb = (uint64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x < 2 * kWave) {   // (one wave per end: 64 probes a step, 4 round trips where a bisection takes 22)
        const unsigned wv = threadIdx.x / kWave;
        const uint64_t tt = wv == 0 ? tb : min(tb + blockDim.x, nnz) - 1;
        const uint64_t k = wave_upper_bound(a_colptr, k0, k1 + 1, e0 + (int64_t)tt) - 1;
        if (threadIdx.x % kWave == 0) krange[wv] = k;
    }
    __syncthreads();
    const uint64_t t = tb + threadIdx.x;
    if (t >= nnz) return;
    const uint64_t k = upper_bound_dev(a_colptr, krange[0], krange[1] + 1, e0 + (int64_t)t) - 1;
    const uint32_t len = (uint32_t)(b_rowptr[k + 1] - b_rowptr[k]);
    if (a_vals_raw) {
        const uint64_t e = (uint64_t)e0 + t;
        const uint32_t v0 = a_vals_raw[e * vwords], v1 = vwords == 2 ? a_vals_raw[e * vwords + 1] : 0u;
        reinterpret_cast<uint4 *>(w)[t] = make_uint4(len, (uint32_t)b_rowptr[k], v0, v1);
    } else if (bs) reinterpret_cast<uint2 *>(w)[t] = make_uint2(len, (uint32_t)b_rowptr[k]);   // pairs: `w` has 2 * nnz words then
    else w[t] = len;
}

}  // namespace osp
