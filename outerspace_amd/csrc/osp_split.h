// osp_split.h -- long rows: by column range, so that the LDS tile kernel can merge them too.
//
// A row whose partial products exceed one LDS tile (thousands of chunks' worth in skewed matrices)
// reaches the tile kernel cut into column ranges ("segments") in a second buffer; inside a segment
// the staging order (ascending k) is preserved, so the merge still sums equal keys in the oracle's
// order.  The segments then look like short rows: the ordinary tile planner packs them and
// merge_tiles_kernel merges them; concatenated in segment order they are the row's sorted result.
// Three ways there (split_params_kernel decides per row):
//   direct   the multiply phase writes the row range by range itself, following a plan made before it
//            (direct_plan_kernel: exact histogram of the row's columns, ranges of at most one tile,
//            one cell per chunk and range) -- no pass over the row's records at all;
//   split    one workgroup moves the row with ONE stable segmented counting-sort pass on the top b
//            bits of the column (split_row_kernel; rows without a chunk table: the parts-merging
//            entry points, or OSP_DIRECT=0);
//   stretch  rows too long for one workgroup: one workgroup per stretch, device-wide scan.
// (Reference for the operation being implemented: deduplicateCOO, SimSpGEMM.cpp:519-535.  The
// reference simply sorts everything; there is no counterpart of this file.)
#pragma once
#include "osp_kernels.h"

namespace osp {

#ifndef OSP_SPLIT_THREADS
#define OSP_SPLIT_THREADS 256
#endif
constexpr int kSplitThreads = OSP_SPLIT_THREADS;  // workgroup of the stretch split (count and scatter).  (1024 threads with rounds
                                                  // of 16384 entries -- runs four times as long, one workgroup per CU -- measured: Graph500
                                                  // scale 20 500 against 505 ms, scale 22 3.97 against 4.06 s; not worth 144 KB of LDS)
#ifndef OSP_SPLIT_ROW_THREADS
#define OSP_SPLIT_ROW_THREADS 1024
#endif
#ifndef OSP_SPLIT_ROW_PRE
#define OSP_SPLIT_ROW_PRE 1
#endif
constexpr int kSplitRowThreads = OSP_SPLIT_ROW_THREADS;  // split_row_kernel's workgroup: 16 waves on one row (see the notes at the kernel)
#ifndef OSP_SPLIT_ROW_STRETCH
#define OSP_SPLIT_ROW_STRETCH 4096
#endif
constexpr int kSplitRowStretch = OSP_SPLIT_ROW_STRETCH;  // ... of split_row_kernel (one workgroup per row)
#ifndef OSP_SPLIT_STRETCH
#define OSP_SPLIT_STRETCH 4096
#endif
constexpr int kSplitStretch = OSP_SPLIT_STRETCH;   // entries a workgroup of the stretch split holds in registers at a time (one round)
#ifndef OSP_SPLIT_JOB_ROUNDS
#define OSP_SPLIT_JOB_ROUNDS 8
#endif
constexpr int kSplitJob = OSP_SPLIT_JOB_ROUNDS * kSplitStretch;  // entries of one long row handled by one workgroup of the stretch split:
                                              // eight rounds share ONE histogram column (a 4096-cell column per 4096
                                              // entries was a third of the split's traffic on Graph500 inputs)
constexpr int kSplitMaxBits = 12;     // at most 4096 segments per row
#ifndef OSP_SPLIT_TARGET
#define OSP_SPLIT_TARGET 256
#endif
constexpr int kSplitTarget = OSP_SPLIT_TARGET;     // aim for segments of about this many entries
#ifndef OSP_SPLIT_ROW_BITS
#define OSP_SPLIT_ROW_BITS 9
#endif
constexpr int kSplitRowBits = OSP_SPLIT_ROW_BITS;      // rows of at most 2^9 segments (<= 128K entries) are split by ONE workgroup (8: +0.8 %, 10: slower)
constexpr uint64_t kSplitRowMax = (uint64_t)kSplitTarget << kSplitRowBits;  // longer rows: one workgroup per stretch
#ifndef OSP_DIRECT_FINE_BITS
#define OSP_DIRECT_FINE_BITS 9
#endif
constexpr int kDirectFineBits = OSP_DIRECT_FINE_BITS;  // fine bins of direct_plan_kernel: at most 2^this per row (LDS of the planner)

// How a long row reaches the tile kernel (hmode):
//   kModeSplitRow  one workgroup moves it into 2^b column ranges of the second buffer (split_row_kernel)
//   kModeStretch   too long for that: one workgroup per stretch, offsets from a device-wide scan (split_count / split_scatter)
//   kModeDirect    the multiply phase writes it into the second buffer range by range (direct_plan_kernel + store_direct,
//                  osp_kernels.h): no pass over its records at all
constexpr uint8_t kModeSplitRow = 0, kModeStretch = 1, kModeDirect = 2;
#ifndef OSP_DIRECT_THREADS
#define OSP_DIRECT_THREADS 256
#endif
constexpr int kDirectThreads = OSP_DIRECT_THREADS;
constexpr int kDirectCells = 4096;     // LDS words of direct_plan_kernel's (chunk, range) cells: counts and in-chunk starts
constexpr uint64_t kDirectDenseMax = 1ull << 20;   // longest dense row (ranges capped at the accumulators' width) written directly
constexpr int kDirectMaxRanges = 255;  // ranges per direct row (a byte per fine bin names the range)
constexpr uint64_t kDirectRowCells = 1ull << 19;  // (chunk, range) cells of ONE direct row at most
// per long row h: b = number of split bits, the mode, and the sizes that get scanned.
// rowfirst != nullptr: the row's chunks are known (first chunk of every row in (row, k) order), so rows of at most
// direct_max partial products that one workgroup could split are planned as direct rows instead: nseg = an upper bound
// of their ranges (consecutive ranges of the greedy grouping together exceed a tile, hence <= 2U/cap + 1 of them),
// ncell = words of their block in the cell array.
// hub_b != 0: the rows that are left to the stretch split are HUB rows instead -- written by the multiply into 2^hub_b uniform
// column blocks (hub_plan_kernel below); they keep the stretch rows' bookkeeping (jobs of kSplitJob products, one histogram
// column per job), only with the same number of blocks for every such row: the run table of B is made for ONE block width.
__global__ void split_params_kernel(const uint32_t *rows, uint32_t nheavy, const uint64_t *row_off, int colbits,
                                    uint64_t row_max, int bits_cap, const uint32_t *rowfirst, uint64_t direct_max, uint32_t cap,
                                    uint8_t *hbits, uint8_t *hmode, uint32_t *nstretch, uint32_t *nseg, uint64_t *nhist, uint64_t *ncell,
                                    int hub_b = 0, int direct_fine = 0, uint64_t *nrund = nullptr) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nheavy) return;
    const uint64_t U = row_off[rows[h] + 1] - row_off[rows[h]];
    const uint64_t want = (U + kSplitTarget - 1) / kSplitTarget;
    int b = 1;
    while (b < kSplitMaxBits && (1ull << b) < want) b++;
    b = min(b, min(colbits, bits_cap));
    const bool big = U > row_max || b > kSplitRowBits;
    // A row whose bins were capped (a "dense" row: more than one product per 8 columns: kSplitTarget products per range of 2^kDenseBits columns) has at most 2^b ranges, each as wide as
    // a dense accumulator, whatever its length: the multiply can write it by range directly although it is longer than what
    // one workgroup splits -- up to kDirectDenseMax products, which one workgroup of the planner walks in a few milliseconds.
    const bool capped = (1ull << b) < want;
    const uint64_t nranges = min(2 * U / cap + 2, (1ull << b) + 1);
    const uint64_t nc = rowfirst != nullptr ? (uint64_t)(rowfirst[rows[h] + 1] - rowfirst[rows[h]]) : 0ull;
    const uint64_t dmax = (capped && direct_max) ? max(direct_max, kDirectDenseMax) : direct_max;
    // (a row of very many tiny chunks -- nc * nranges cells -- would keep ONE workgroup of the planner busy for milliseconds,
    // block of chunks after block of chunks: such a row is cheaper to split)
    const bool direct = rowfirst != nullptr && b <= kSplitRowBits && (!big || capped) && U <= dmax && nranges <= (uint64_t)kDirectMaxRanges &&
                        nc * nranges <= kDirectRowCells;
    const uint32_t ns = (big && !direct) ? (uint32_t)((U + kSplitJob - 1) / kSplitJob) : 0u;  // 0 stretches = one-workgroup row
    if (ns && hub_b) b = hub_b;
    // A direct row's fine bins only serve its planner (the ranges are groups of bins): bins of kSplitTarget products -- the split's
    // segment size -- leave a range half a bin short of a tile on average (tiles 92 % full on R-MAT-22 "mild"); 2^direct_fine
    // times as many bins (up to the planner's 512) fill the tiles better: fewer tiles, longer runs, fewer cells.
    if (direct && !capped) b = min(b + direct_fine, min(colbits, kDirectFineBits));
    hbits[h] = (uint8_t)b;
    hmode[h] = direct ? kModeDirect : big ? kModeStretch : kModeSplitRow;
    nstretch[h] = ns;
    nseg[h] = direct ? (uint32_t)nranges : 1u << b;
    nhist[h] = (uint64_t)ns << b;
    ncell[h] = direct ? ((1ull << b) + 3) / 4 + nc * nranges : 0ull;
    // run descriptors of a gathered row: one per non-empty (chunk, range) cell, never more than the row has products
    if (nrund) nrund[h] = direct ? min(nc * nranges, U) : 0ull;
}
struct HeavyLenIf {   // partial products of long row h if it has mode `mode` (else 0): how much each path handles
    const uint32_t *rows;
    const uint64_t *row_off;
    const uint8_t *hmode;
    uint8_t mode;
    __device__ uint64_t operator()(uint64_t h) const { return hmode[h] == mode ? row_off[rows[h] + 1] - row_off[rows[h]] : 0ull; }
};
struct LoadU32As64 {
    const uint32_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};

// workgroup -> (long row h, stretch st)
struct SplitJob {
    uint32_t h, st, b, nst;
    uint64_t beg, end;   // entry range in the staging buffer (panel-relative)
    uint64_t hbase;      // first histogram cell of row h
};
__device__ __forceinline__ SplitJob split_job(const uint32_t *rows, uint32_t nheavy, const uint64_t *blkbase,
                                              const uint64_t *hbase, const uint8_t *hbits, const uint32_t *nstretch,
                                              const uint64_t *row_off, uint64_t base) {
    SplitJob j;
    j.h = (uint32_t)(upper_bound_dev(blkbase, 0, (uint64_t)nheavy + 1, (uint64_t)blockIdx.x) - 1);
    j.st = (uint32_t)(blockIdx.x - blkbase[j.h]);
    j.b = hbits[j.h];
    j.nst = nstretch[j.h];
    const uint64_t s = row_off[rows[j.h]] - base, e = row_off[rows[j.h] + 1] - base;
    j.beg = s + (uint64_t)j.st * kSplitJob;
    j.end = min(j.beg + (uint64_t)kSplitJob, e);
    j.hbase = hbase[j.h];
    return j;
}

// histogram of one stretch over the row's segments -> ghist[hbase + seg * nst + st]
__global__ __launch_bounds__(kSplitThreads) void split_count_kernel(
    const uint32_t *rows, uint32_t nheavy, const uint64_t *blkbase, const uint64_t *hbase, const uint8_t *hbits,
    const uint32_t *nstretch, const uint64_t *row_off, uint64_t base, int colbits, const char *stage, uint32_t rec_bytes,
    uint32_t *ghist) {
    __shared__ uint32_t hist[1 << kSplitMaxBits];
    const SplitJob j = split_job(rows, nheavy, blkbase, hbase, hbits, nstretch, row_off, base);
    const uint32_t nseg = 1u << j.b;
    for (uint32_t d = threadIdx.x; d < nseg; d += kSplitThreads) hist[d] = 0;
    __syncthreads();
    const int sh = colbits - (int)j.b;
    for (uint64_t i = j.beg + threadIdx.x; i < j.end; i += kSplitThreads)
        atomicAdd(&hist[*(const uint32_t *)(stage + i * rec_bytes) >> sh], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nseg; d += kSplitThreads) ghist[j.hbase + (uint64_t)d * j.nst + j.st] = hist[d];
}

// Lanes of this wave with the same `bits`-bit digit: rank among them (lower lanes first) and group size.
__device__ __forceinline__ void wave_match_bits(unsigned digit, int bits, bool valid, unsigned &rank, unsigned &count) {
    const uint64_t vm = __ballot(valid);
    uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
    for (int b = 0; b < bits; b++) {
        const uint32_t bit = (digit >> b) & 1u;
        const uint64_t m = __ballot(bit != 0);
        const uint32_t sbm = 0u - bit;
        plo &= ~((uint32_t)m ^ sbm);
        phi &= ~((uint32_t)(m >> 32) ^ sbm);
    }
    rank = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    count = __popc(plo) + __popc(phi);
}

// stable scatter of one stretch into the row's segments; goffs = exclusive scan of ghist
// (Measured and not kept: the per-segment bookkeeping through 8- and 16-byte LDS accesses, four segments per thread -- a
// quarter of its LDS instructions: no change at 512 or 2048 segments per row (500.4 against 500.5 ms, 4.02 against 4.03 s).
// The kernel waits for its scattered 24..96-byte writes, not for LDS.)
template <class T, bool RA>
__global__ __launch_bounds__(kSplitThreads) void split_scatter_kernel(
    const uint32_t *rows, uint32_t nheavy, const uint64_t *blkbase, const uint64_t *hbase, const uint8_t *hbits,
    const uint32_t *nstretch, const uint64_t *row_off, uint64_t base, int colbits, const Part<T> *stage,
    const uint32_t *goffs, const uint64_t *hoff, Part<T> *qstage) {
    constexpr int NW = kSplitThreads / kWave;
    constexpr int ITERS = kSplitStretch / kSplitThreads;  // 16 wave iterations per wave span
    __shared__ alignas(8) uint16_t cnt[NW][1 << kSplitMaxBits];  // (pairs of counters are also addressed as 32-bit words)
    __shared__ uint32_t boff[1 << kSplitMaxBits];
    const SplitJob j = split_job(rows, nheavy, blkbase, hbase, hbits, nstretch, row_off, base);
    const uint32_t nseg = 1u << j.b;
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t qrow = (uint32_t)hoff[j.h] - goffs[j.hbase];  // the scan covers the multi-workgroup rows only
    for (uint32_t d = threadIdx.x; d < nseg; d += kSplitThreads) {
        boff[d] = qrow + goffs[j.hbase + (uint64_t)d * j.nst + j.st];
#pragma unroll
        for (int ww = 0; ww < NW; ww++) cnt[ww][d] = 0;
    }
    __syncthreads();
    const int sh = colbits - (int)j.b;
    // the job's entries in rounds of kSplitStretch; boff[d] runs along (it always points behind what the earlier rounds
    // put into segment d)
    for (uint64_t sb = j.beg; sb < j.end; sb += kSplitStretch) {
        const uint64_t se = min(sb + (uint64_t)kSplitStretch, j.end);
        // each wave owns a contiguous quarter of the round: earlier waves = earlier entries (stable)
        const uint64_t wbeg = sb + (uint64_t)w * (kSplitStretch / NW);
        uint32_t rk[ITERS];
        PartWords<T> rec[ITERS];  // raw records: all loads in flight together
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t i = wbeg + (uint64_t)it * kWave + lane;
            rec[it] = load_part_words(&stage[i < se ? i : sb]);  // branch-free: lanes past the end re-read the first record
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t i = wbeg + (uint64_t)it * kWave + lane;
            const bool valid = i < se;
            const unsigned d = rec[it].col() >> sh;
            rk[it] = 0;
            if constexpr (RA) {
                // stable rank by LDS atomic on the wave's packed 16-bit counter (see merge_tiles_kernel)
                const unsigned half = 16u * (d & 1u);
                if (valid) rk[it] = (atomicAdd(&reinterpret_cast<uint32_t *>(cnt[w])[d >> 1], 1u << half) >> half) & 0xffffu;
            } else {
                unsigned r, c;
                wave_match_bits(d, (int)j.b, valid, r, c);
                if (valid) {
                    const uint32_t cur = cnt[w][d];
                    rk[it] = cur + r;
                    if (r == 0) cnt[w][d] = (uint16_t)(cur + c);
                }
            }
        }
        __syncthreads();
        // per segment: the waves' exclusive offsets, stored RELATIVE TO THE END of the round's entries of the segment
        // (a small negative number in 16 bits), and boff moved to that end -- no extra counter row for the totals
        for (uint32_t d = threadIdx.x; d < nseg; d += kSplitThreads) {
            uint32_t total = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) total += cnt[ww][d];
            uint32_t run = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { const uint32_t c = cnt[ww][d]; cnt[ww][d] = (uint16_t)(run - total); run += c; }
            boff[d] += total;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t i = wbeg + (uint64_t)it * kWave + lane;
            if (i < se) {
                const unsigned d = rec[it].col() >> sh;
                const uint32_t dst = boff[d] + (uint32_t)(int32_t)(int16_t)cnt[w][d] + rk[it];
                store_part_words(&qstage[dst], rec[it]);
            }
        }
        __syncthreads();  // the scatter has read the counters
        for (uint32_t d = threadIdx.x; d < nseg; d += kSplitThreads) {
#pragma unroll
            for (int ww = 0; ww < NW; ww++) cnt[ww][d] = 0;
        }
        __syncthreads();
    }
}

// segment v of the split = "virtual row": its offset in the second buffer
// ... and the columns it covers, [vcol0[v], vcol1[v]) (the segments of a direct row: written by its planner)
__global__ void split_vrows_kernel(uint32_t nheavy, const uint64_t *vbase, const uint64_t *hbase, const uint32_t *nstretch,
                                   const uint8_t *hbits, const uint8_t *hmode, int colbits,
                                   const uint32_t *goffs, const uint64_t *hoff, uint64_t nvirt, uint64_t nh_total,
                                   uint64_t *vrow_off, uint8_t *vfirst, uint32_t *vcol0, uint32_t *vcol1) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nvirt) return;
    if (v == nvirt) { vrow_off[v] = nh_total; vfirst[v] = 1; return; }
    const uint32_t h = (uint32_t)(upper_bound_dev(vbase, 0, (uint64_t)nheavy + 1, v) - 1);
    const uint64_t d = v - vbase[h];
    vfirst[v] = d == 0;  // first segment of a long row: a tile must start here
    if (hmode[h] == kModeDirect) return;
    const int sh = colbits - (int)hbits[h];
    vcol0[v] = (uint32_t)(d << sh);
    vcol1[v] = (uint32_t)min((d + 1) << sh, (uint64_t)0xffffffffu);  // (columns are below 2^32 - 1)
    if (nstretch[h] == 0) return;  // one-workgroup row: split_row_kernel wrote its segment offsets
    vrow_off[v] = hoff[h] + (goffs[hbase[h] + d * nstretch[h]] - goffs[hbase[h]]);
}

// ---- over-long segments of hub rows: dense accumulation ------------------------------------------------------------------
// A segment that still exceeds a merge tile after the split is, in a hub row, a NARROW column range fed by thousands of
// products: a row of U partial products is cut into up to 4096 ranges, so at Graph500 skew the range of a segment is a few
// hundred columns while the segment holds thousands of entries (hub row x hub column).  Sorting that is the wrong tool
// (one 4096-entry LDS tile per segment, or five global radix passes for the "piles" beyond: a fifth of the Graph500
// product's time).  A wave takes such a segment and keeps ONE accumulator per column of its range in LDS: entries are
// consumed 64 at a time in staging order (= ascending k); lanes that hold the same column inside one group are ranked
// (ballot matching) and added in rank order, one round per rank, so every column's sum is formed in exactly the order the
// sort-and-sum would use -- same bits -- and different columns proceed in parallel.  A bitmap remembers which columns
// occurred (a sum that cancels to zero is still an entry).  The compacted (column, sum) pairs go back over the segment's
// own beginning, like the other paths for over-long segments.
#ifndef OSP_DENSE_BITS
#define OSP_DENSE_BITS 11
#endif
constexpr int kDenseBits = OSP_DENSE_BITS;  // segments whose column range is at most 2^kDenseBits columns
// waves (= segments in progress) per workgroup: what fits 160 KB of LDS with f64 accumulators, at most four
constexpr int kDenseWaves = kDenseBits <= 12 ? 4 : (kDenseBits == 13 ? 2 : 1);
struct SegDenseFlag {
    const uint32_t *list;   // over-long segments (virtual rows)
    const uint32_t *vcol0, *vcol1;  // the columns every segment covers
    int enabled;
    __device__ uint32_t operator()(uint64_t t) const {
        if (!enabled) return 0u;
        const uint32_t v = list[t];
        return (vcol1[v] - vcol0[v]) <= (1u << kDenseBits) ? 1u : 0u;
    }
};
// Does one ds_add_f64 apply the lanes that hit the same address in ascending lane order?  Every wave adds 64 values of
// wildly different magnitude (so that any other order changes the bits) to a few accumulators, once by the atomic and once
// lane by lane, and compares the bits.
template <class T>
__global__ __launch_bounds__(256) void fadd_order_selftest_kernel(uint32_t *bad) {
    __shared__ T acc[4][8], ref[4][8];
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    uint32_t x = 0x9E3779B9u * (blockIdx.x * 256u + tid + 1u), nbad = 0;
    for (int r = 0; r < 64; r++) {
        if (lane < 8) { acc[w][lane] = T(-0.0); ref[w][lane] = T(-0.0); }
        __builtin_amdgcn_wave_barrier();
        for (int it = 0; it < 4; it++) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const unsigned c = (x >> 9) & ((r & 1) ? 1u : 7u);
            const bool valid = ((x >> 27) & 7u) != 0;
            // mantissa from x, exponent spread over 2^-40 .. 2^40 (2^-12 .. 2^12 in f32), both signs; every eighth round:
            // subnormals and their neighbours (exponent fields 0..2), which the atomic must not flush
            T v;
            if constexpr (sizeof(T) == 8) {
                const uint64_t expo = (r & 7) == 7 ? (uint64_t)((x >> 3) % 3u) : (uint64_t)(1023 - 40 + ((x >> 3) % 81u));
                v = __longlong_as_double((long long)(((uint64_t)(x & 0x80000000u) << 32) | (expo << 52) |
                                                     ((uint64_t)x * 0x9E3779B97F4A7ull & 0xFFFFFFFFFFFFFull)));
            } else {
                const uint32_t expo = (r & 7) == 7 ? (x >> 3) % 3u : 127u - 12u + ((x >> 3) % 25u);
                v = __uint_as_float((x & 0x80000000u) | (expo << 23) | ((x * 0x9E3779B9u) & 0x7FFFFFu));
            }
            if (valid) __hip_atomic_fetch_add(&acc[w][c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (unsigned l = 0; l < kWave; l++) {   // the reference: lane by lane
                if (lane == l && valid) ref[w][c] += v;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 8) {
            if constexpr (sizeof(T) == 8) nbad += __double_as_longlong(acc[w][lane]) != __double_as_longlong(ref[w][lane]);
            else nbad += __float_as_uint(acc[w][lane]) != __float_as_uint(ref[w][lane]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
}
// list[t] -> yes[scan[t]] or no[t - scan[t]]; scan = exclusive scan of the flags
template <class F>
__global__ void seg_split_list_kernel(F f, const uint32_t *scan, uint32_t n, uint32_t *yes, uint32_t *no) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (f(t)) yes[scan[t]] = f.list[t]; else no[t - scan[t]] = f.list[t];
}
// FA: the additions are ONE LDS floating-point atomic per group of 64 entries instead of ballot ranks and
// rounds.  Lanes of one ds_add_f64 / ds_add_f32 that hit the same accumulator are applied in ascending lane order -- the same property
// of the LDS atomic unit the stable ranks rest on (osp_kernels.h), equally undocumented, so fadd_order_selftest_kernel
// checks it (and that subnormals are not flushed) for each value type on the device when the context is created and the ballot form below stays as the fallback -- and successive
// instructions of a wave execute in order: every column's sum is formed in staging order.  An accumulator starts as
// -0.0, the one value with (-0.0) + x == x for every x: "the sum starts AS the first entry" without telling first from
// later entries.
template <class T, bool FA = false>
__global__ __launch_bounds__(kDenseWaves * kWave) void dense_segment_kernel(const uint32_t *list, uint32_t nlist, const uint64_t *vrow_off,
                                                                          const uint32_t *vcol0, const uint32_t *vcol1,
                                                                          Part<T> *qstage, uint32_t *seg_nnz) {
    constexpr int NW = kDenseWaves, R_MAX = 1 << kDenseBits;
    __shared__ T acc[NW][R_MAX];
    __shared__ uint32_t seen[NW][R_MAX / 32];
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t idx = blockIdx.x * NW + w;
    if (idx >= nlist) return;  // (no workgroup barrier below: the waves are independent)
    const uint32_t v = list[idx];
    const uint32_t cbase = vcol0[v], R = vcol1[v] - cbase;   // (<= 2^kDenseBits: SegDenseFlag)
    int sh = 0;                                              // bits that hold a column relative to cbase
    while ((1u << sh) < R) sh++;
    const uint64_t s0 = vrow_off[v], m = vrow_off[v + 1] - s0;
    for (uint32_t c = lane; c < (R + 31) / 32; c += kWave) seen[w][c] = 0u;
    constexpr bool ATOMIC = FA;
    if constexpr (ATOMIC) {
        for (uint32_t c = lane; c < R; c += kWave) acc[w][c] = T(-0.0);
    }
    __builtin_amdgcn_wave_barrier();
    Part<T> *seg = qstage + s0;
    if constexpr (ATOMIC) {
#ifndef OSP_DENSE_UNROLL
#define OSP_DENSE_UNROLL 16   // (4 until late in round 3; two workgroups of four waves per CU live on what each wave keeps in flight: Graph500 scale 18 ef 64 499 / 475 / 467 ms with 4 / 8 / 16)
#endif
        constexpr int UNROLL = OSP_DENSE_UNROLL;  // groups of loads in flight; the atomics are issued group by group, in order
        for (uint64_t i0 = 0; i0 < m; i0 += (uint64_t)UNROLL * kWave) {
            PartWords<T> rec[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint64_t i = i0 + (uint64_t)u * kWave + lane;
                rec[u] = load_part_words(&seg[i < m ? i : 0]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint64_t i = i0 + (uint64_t)u * kWave + lane;
                if (i < m) {
                    const uint32_t rel = rec[u].col() - cbase;
                    __hip_atomic_fetch_add(&acc[w][rel], rec[u].val(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    atomicOr(&seen[w][rel >> 5], 1u << (rel & 31u));
                }
            }
        }
    }
    for (uint64_t i0 = 0; !ATOMIC && i0 < m; i0 += kWave) {
        const uint64_t i = i0 + lane;
        const bool valid = i < m;
        const PartWords<T> rec = load_part_words(&seg[valid ? i : 0]);
        const uint32_t rel = valid ? rec.col() - cbase : 0u;
        unsigned rk, cnt;
        wave_match_bits(rel, sh, valid, rk, cnt);
        // round r: the lanes whose entry is the r-th of its column inside this group -- distinct columns, no conflict
        for (unsigned r = 0; __ballot(valid && rk >= r) != 0; r++) {
            if (valid && rk == r) {
                const uint32_t bit = 1u << (rel & 31u);
                const uint32_t old = seen[w][rel >> 5];
                // a column's sum STARTS as its first entry (0 + (-0.0) would lose the sign of a lone negative zero)
                if (old & bit) acc[w][rel] += rec.val();
                else { acc[w][rel] = rec.val(); atomicOr(&seen[w][rel >> 5], bit); }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // compact: columns in ascending order, back over the segment's own beginning (all of it has been consumed)
    uint32_t out = 0;
    for (uint32_t c0 = 0; c0 < R; c0 += kWave) {
        const uint32_t c = c0 + lane;
        const bool on = c < R && ((seen[w][c >> 5] >> (c & 31u)) & 1u);
        const uint64_t mask = __ballot(on);
        if (on) {
            const uint32_t pos = out + (uint32_t)__popcll(mask & lanemask_lt());
            seg[pos] = Part<T>{cbase + c, acc[w][c]};
        }
        out += (uint32_t)__popcll(mask);
    }
    if (lane == 0) seg_nnz[v] = out;
}

// One workgroup splits one long row of at most kSplitRowMax entries: histogram over its segments, scan, stable
// scatter -- the row is read twice, the second time from cache.  No global histogram, no device-wide scan.
// Two variants were built, measured on R-MAT-22 (28 GB of long-row records per launch, 18.6 ms as it stands) and NOT kept:
//   * segment counts from column-block tables of B's rows (257 prefix counts per row of B, built once per product) instead
//     of the histogram pass: 17.4 ms, but 4 ms per product to build the table -- the histogram pass costs hardly more than
//     a millisecond of the kernel, because it leaves the row in cache for the scatter pass;
//   * the scatter staged through LDS (a round's records put in segment order inside the workgroup, then streamed out with
//     consecutive threads on consecutive records): 18.9 ms -- the direct 12-byte stores are not what limits it;
//   * more waves per CU (workgroups of 256 threads holding 2048 / 1024 entries per round: 62 / 42 registers, 8 waves per
//     SIMD): 20.0 / 21.0 ms -- SLOWER.  The kernel waits on memory (78 % of its wave cycles parked, 6 % issuing), but what
//     it waits for is the second read of rows that more rows in flight have pushed out of L2 (FETCH_SIZE: every row read
//     twice).  Hence the opposite: FEWER rows in flight, each finished sooner -- workgroups of 1024 threads (4 records per
//     thread and round): 17.0 ms; 512 threads 17.7; one such workgroup per CU instead of two 16.8 (not kept: within noise);
//     rounds of 8192 / 16384 entries with 1024 threads 21.5.  The first round's records kept in registers from the
//     histogram to the scatter: 16.4 ms (FETCH_SIZE 27.8 -> 22.0 GB per launch); the first TWO rounds'
//     (-DOSP_SPLIT_ROW_PRE=2, 62 registers): 16.5 ms -- no further gain, the read volume is no longer what it waits for.
template <class T, bool RA>
__global__ __launch_bounds__(kSplitRowThreads) void split_row_kernel(
    const uint32_t *rows, uint32_t nheavy, const uint8_t *hbits, const uint8_t *hmode, const uint64_t *vbase,
    const uint64_t *hoff, const uint64_t *row_off, uint64_t base, int colbits, const Part<T> *stage, Part<T> *qstage,
    uint64_t *vrow_off) {
    constexpr int NW = kSplitRowThreads / kWave;
    constexpr int ITERS = kSplitRowStretch / kSplitRowThreads;
    constexpr int NSEG = 1 << kSplitRowBits;
    static_assert(NSEG <= kSplitRowThreads, "one thread per segment in the scan");
    __shared__ alignas(8) uint16_t cnt[NW + 1][NSEG];
    __shared__ uint32_t segoff[NSEG];  // histogram, then running offset of every segment
    __shared__ uint32_t scratch[NW + 1];
    const uint32_t h = blockIdx.x;
    if (h >= nheavy || hmode[h] != kModeSplitRow) return;
    const unsigned lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t b = hbits[h], nseg = 1u << b;
    const int sh = colbits - (int)b;
    const uint64_t beg = row_off[rows[h]] - base, end = row_off[rows[h] + 1] - base;
    const uint64_t qbase = hoff[h];
    for (uint32_t d = threadIdx.x; d < nseg; d += kSplitRowThreads) segoff[d] = 0;
    // the records of the FIRST round are loaded whole and kept for its scatter (a third of the long-row records sit in the
    // first 4096 entries of their row: read once instead of twice); the rest of the row is counted from its column words
    PartWords<T> rec[ITERS];
#if OSP_SPLIT_ROW_PRE >= 2
    PartWords<T> rec1[ITERS];  // ... and the second round's (rows of up to 8192 entries are then read once)
#endif
    {
        const uint64_t wbeg0 = beg + (uint64_t)w * (kSplitRowStretch / NW);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t i = wbeg0 + (uint64_t)it * kWave + lane;
            rec[it] = load_part_words(&stage[i < end ? i : beg]);  // branch-free: lanes past the end re-read the first record
        }
#if OSP_SPLIT_ROW_PRE >= 2
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t i = wbeg0 + kSplitRowStretch + (uint64_t)it * kWave + lane;
            rec1[it] = load_part_words(&stage[i < end ? i : beg]);
        }
#endif
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITERS; it++)
            if (wbeg0 + (uint64_t)it * kWave + lane < end) atomicAdd(&segoff[rec[it].col() >> sh], 1u);
#if OSP_SPLIT_ROW_PRE >= 2
#pragma unroll
        for (int it = 0; it < ITERS; it++)
            if (wbeg0 + kSplitRowStretch + (uint64_t)it * kWave + lane < end) atomicAdd(&segoff[rec1[it].col() >> sh], 1u);
#endif
    }
    for (uint64_t i = beg + (uint64_t)OSP_SPLIT_ROW_PRE * kSplitRowStretch + threadIdx.x; i < end; i += kSplitRowThreads)
        atomicAdd(&segoff[stage[i].col >> sh], 1u);
    __syncthreads();
    {   // exclusive scan of the segment counts -> segment offsets
        const uint32_t c = threadIdx.x < nseg ? segoff[threadIdx.x] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan<uint32_t, kSplitRowThreads>(c, scratch, &total);
        if (threadIdx.x < nseg) { segoff[threadIdx.x] = ex; vrow_off[vbase[h] + threadIdx.x] = qbase + ex; }
    }
    __syncthreads();
    for (uint64_t sb = beg; sb < end; sb += kSplitRowStretch) {
        const uint64_t se = min(sb + (uint64_t)kSplitRowStretch, end);
        for (uint32_t d = threadIdx.x; d < nseg; d += kSplitRowThreads) {
#pragma unroll
            for (int ww = 0; ww < NW; ww++) cnt[ww][d] = 0;
        }
        __syncthreads();  // also orders the segoff update of the previous round before this round's scatter
        const uint64_t wbeg = sb + (uint64_t)w * (kSplitRowStretch / NW);
        uint32_t rk[ITERS];
#if OSP_SPLIT_ROW_PRE >= 2
        if (sb == beg + kSplitRowStretch) {
#pragma unroll
            for (int it = 0; it < ITERS; it++) rec[it] = rec1[it];
        } else
#endif
        if (sb != beg) {
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const uint64_t i = wbeg + (uint64_t)it * kWave + lane;
                rec[it] = load_part_words(&stage[i < se ? i : sb]);
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t i = wbeg + (uint64_t)it * kWave + lane;
            const bool valid = i < se;
            const unsigned d = rec[it].col() >> sh;
            rk[it] = 0;
            if constexpr (RA) {
                const unsigned half = 16u * (d & 1u);
                if (valid) rk[it] = (atomicAdd(&reinterpret_cast<uint32_t *>(cnt[w])[d >> 1], 1u << half) >> half) & 0xffffu;
            } else {
                unsigned r, c;
                wave_match_bits(d, (int)b, valid, r, c);
                if (valid) {
                    const uint32_t cur = cnt[w][d];
                    rk[it] = cur + r;
                    if (r == 0) cnt[w][d] = (uint16_t)(cur + c);
                }
            }
        }
        __syncthreads();
        // per segment: exclusive offsets of the waves inside this stretch; cnt[NW] keeps the stretch's total
        for (uint32_t d = threadIdx.x; d < nseg; d += kSplitRowThreads) {
            uint32_t run = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) { const uint32_t c = cnt[ww][d]; cnt[ww][d] = (uint16_t)run; run += c; }
            cnt[NW][d] = (uint16_t)run;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint64_t i = wbeg + (uint64_t)it * kWave + lane;
            if (i < se) {
                const unsigned d = rec[it].col() >> sh;
                store_part_words(&qstage[qbase + segoff[d] + cnt[w][d] + rk[it]], rec[it]);
            }
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < nseg; d += kSplitRowThreads) segoff[d] += cnt[NW][d];
        // (the zeroing of cnt at the top of the next round touches the same d from the same thread)
    }
}
// ---- direct rows: the plan the multiply phase writes by ------------------------------------------------------------------
// One workgroup per direct row (hmode == kModeDirect).  The row's chunks -- one per non-zero A[i,k], in ascending k -- and
// the B rows they multiply are known from the symbolic phase (the chunk table); nothing of the row has been computed yet.
//   1. exact histogram of the row's product columns over the 2^b fine bins (col >> sh) the split would use: B's column
//      indices are read chunk by chunk (runs of equal bins inside a wave-load are added by their first lane);
//   2. consecutive bins are grouped greedily into RANGES of at most `cap` products (a bin that exceeds `cap` alone is a
//      range of its own: an over-long segment for the paths that handle those) -> the row's segments: offsets in the second
//      buffer, column bounds, and the byte table bin -> range;
//   3. the chunks are walked again, a block of them at a time: per (chunk, range) the number of its entries (cells in
//      LDS), row prefixes (where the range starts inside the chunk) and column prefixes carried from block to block
//      (where the chunk's run starts inside the range) -> the cell the multiply adds an entry's index to.
// Ranges of one row stay in column order, runs inside a range in chunk order (ascending k), entries inside a run in
// column order: the layout a stable split of the k-ordered row would give.
// head lanes of the runs of equal `key` among the valid lanes of a wave-load, and each run's length
// (the planner that calls this is bound by its VALU instructions: ballots straight into SGPRs, the lane's "lanes above me"
// mask from the caller -- it does not change between calls)
__device__ __forceinline__ uint64_t lanes_above_mask() { return lane_id() == 63 ? 0ull : (~0ull << (lane_id() + 1)); }
__device__ __forceinline__ bool wave_run_head(uint32_t key, bool valid, uint32_t &runlen, uint64_t above) {
    const unsigned lane = lane_id();
    const uint32_t prev = wave_shr1(key);
    const bool head = valid && (lane == 0 || key != prev);
    const uint64_t heads = __builtin_amdgcn_ballot_w64(head), vm = __builtin_amdgcn_ballot_w64(valid);
    const uint64_t rest = heads & above;
    const uint32_t next = rest ? (uint32_t)__builtin_ctzll(rest) : (uint32_t)__popcll(vm);  // valid lanes are a prefix
    runlen = next - lane;
    return head;
}
// The row's chunks are taken a block at a time: their B rows and entry counts go to LDS (one coalesced load), a scan
// turns the counts into a flat numbering of the block's entries, and the workgroup's threads walk that numbering -- four
// wave-loads of B's column indices in flight per wave, whatever the chunk lengths (a wave per chunk waited for two
// dependent loads per chunk: 24 ms per launch on R-MAT-22 instead of 3).
#ifndef OSP_DIRECT_UNR
#define OSP_DIRECT_UNR 8
#endif
// The planner is bound by its VALU instructions (88 % VALU utilisation by the SQ counters, MEASUREMENTS.md 1.6), and merging
// the runs of equal (chunk, range) keys of a wave-load before the LDS atomic (wave_run_head) is a dozen of them per load --
// but it pays: with one atomic per entry instead (0) the kernel takes 18.3 ms per launch against 10.4, same-address LDS
// atomics being resolved one after the other.
#ifndef OSP_PLAN_RUNS
#define OSP_PLAN_RUNS 1
#endif
#ifndef OSP_DIRECT_CHUNK_BLOCK
#define OSP_DIRECT_CHUNK_BLOCK 512
#endif
#ifndef OSP_DIRECT_CELLS_LDS
#define OSP_DIRECT_CELLS_LDS 1024
#endif
constexpr int kDirectChunkBlock = OSP_DIRECT_CHUNK_BLOCK;    // chunks of a row whose descriptors sit in LDS at a time
// first index c in [0, n) with cst[c + 1] > i   (cst ascending, cst[n] > i)
__device__ __forceinline__ uint32_t direct_find_chunk(const uint32_t *cst, uint32_t n, uint32_t i) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cst[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
// Phase timing of the planner (-DOSP_PLAN_PROF, a development build: OSP_VERBOSE prints the shares after every product):
// thread 0 of every workgroup adds the cycles between marks to osp_plan_prof[phase].  Compiled out of the library.
#ifdef OSP_PLAN_PROF
__device__ unsigned long long osp_plan_prof[8];
#define OSP_PLAN_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&osp_plan_prof[k], now_ - prof_t); prof_t = now_; } } while (0)
#define OSP_PLAN_DECL unsigned long long prof_t = clock64();
#else
#define OSP_PLAN_MARK(k)
#define OSP_PLAN_DECL
#endif
// GATHERED rows (round 5; osp_kernels.h, "gathered rows"): a direct row none of whose ranges exceeds a tile is not written by
// the multiply at all.  Step 3 then leaves, instead of cells, one RunDesc per non-empty (chunk, range) cell in the order
// (range, chunk) -- where the run would start in the second buffer, where it starts in B, the chunk's A value -- and the
// bounds of every segment's descriptors (vrun_off / vrun_end); the row's chunks get kChunkSkip.  The order needs the number
// of non-empty cells of every range before the first descriptor is written: a row whose cells fit LDS at once (nearly all)
// counts them there; a row of several blocks of chunks leaves the blocks' counts in its (otherwise unused) cell block of
// `cells` and reads them back, so B's columns are still walked twice, not three times.
// rowruns[h]: the row's number of runs, kNoRuns for a row that is not gathered (statistics: gather_stats_kernel).
struct GatherPlan {
    const uint64_t *rdbase = nullptr;   // first descriptor of every long row's block in the run table (null: no row is gathered)
    uint32_t *vrun_off = nullptr, *vrun_end = nullptr;   // per segment (virtual row): its descriptors
    uint32_t *rowruns = nullptr;
    uint32_t *nwritten = nullptr;       // += 1 per direct row that is NOT gathered: rows the multiply writes through cells
    uint32_t over = 1;                  // rows with a range that exceeds a tile are gathered too
    uint32_t mark_skipped = 1;          // the chunk offsets exist (a column-major multiply may read them): gathered chunks get kChunkSkip
    uint32_t av_in_order = 0;           // a_vals holds the chunks' A values in (row, k) order (else: indexed through perm)
};
// gstat[0..2] += gathered rows, their partial products, their runs (few workgroups: they end in atomics on three hot words --
// one set per ROW inside the planner made it twice as slow)
__global__ void gather_stats_kernel(const uint32_t *rows, uint32_t nlong, const uint64_t *row_off, const uint32_t *rowruns, unsigned long long *gstat) {
    uint64_t nr = 0, np = 0, nd = 0;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nlong; h += gridDim.x * blockDim.x) {
        const uint32_t r = rowruns[h];
        if (r != kNoRuns) { nr++; nd += r; np += row_off[rows[h] + 1] - row_off[rows[h]]; }
    }
    nr = wave_reduce_sum<uint64_t>(nr); np = wave_reduce_sum<uint64_t>(np); nd = wave_reduce_sum<uint64_t>(nd);
    if (lane_id() == 0 && nr) { atomicAdd(&gstat[0], (unsigned long long)nr); atomicAdd(&gstat[1], (unsigned long long)np); atomicAdd(&gstat[2], (unsigned long long)nd); }
}
// CL: LDS words for the (chunk, range) cells of a block of chunks, twice.  A row of many short chunks and a dozen ranges (the web
// graphs' long rows: 1 125 chunks, 12 ranges) takes its chunks 1024 / 12 = 85 at a time, every block a chain of dependent loads
// and barriers: fine while thousands of rows hide each other's latencies, but a launch of a few hundred rows lasts as long
// as its longest row (0.21 ms of a 1.8 ms product).  Such launches -- all rows resident at once even at two workgroups per CU
// -- run the instantiation with eight times the cells (and the block's A values in an array of their own).
constexpr int kDirectCellsBig = 8 * OSP_DIRECT_CELLS_LDS;
template <class V, int CL = OSP_DIRECT_CELLS_LDS>
__global__ __launch_bounds__(kDirectThreads, (CL > OSP_DIRECT_CELLS_LDS ? 2 : 8)) void direct_plan_kernel(
    const uint32_t *__restrict__ rows, uint32_t nlong, const uint8_t *__restrict__ hmode, const uint8_t *__restrict__ hbits,
    const uint32_t *__restrict__ nseg, const uint64_t *__restrict__ vbase, const uint64_t *__restrict__ hoff,
    const uint64_t *__restrict__ cellbase, const uint64_t *__restrict__ row_off, int colbits, uint32_t cap,
    const uint32_t *__restrict__ rowfirst, const uint64_t *__restrict__ ct_off, const uint32_t *__restrict__ ct_bs,
    const uint32_t *__restrict__ perm, const uint32_t *__restrict__ b_colidx, uint64_t *__restrict__ vrow_off,
    uint32_t *__restrict__ vcol0, uint32_t *__restrict__ vcol1, uint32_t *__restrict__ cells, uint64_t *__restrict__ chunk_off,
    const GatherPlan gp, const V *__restrict__ a_vals, RunDesc<V> *__restrict__ runs) {
    constexpr int NT = kDirectThreads, NFINE = 1 << kDirectFineBits, CBL = kDirectChunkBlock, UNR = OSP_DIRECT_UNR;
    constexpr int kCellsLds = CL;
    constexpr bool kBigCells = CL > OSP_DIRECT_CELLS_LDS;
    __shared__ V avs_own[kBigCells ? CBL : 1];
    __shared__ alignas(8) uint32_t hist[NFINE + 1];     // bin counts, then their exclusive prefix (gathered rows, step 3: the block's A values)
    __shared__ alignas(4) uint16_t nxt[NFINE];          // first bin of the range that follows a range starting at this bin (gathered rows, step 3: runs per range)
    __shared__ uint8_t lut[NFINE];
    __shared__ uint32_t rbin0[kDirectMaxRanges + 2], roff[kDirectMaxRanges + 2], cursor[kDirectMaxRanges + 1];
    __shared__ uint32_t cbs[CBL], cst[CBL + 1];   // per chunk of the block: (B position - first entry number), first entry number
    __shared__ uint32_t cellm[kCellsLds], lsm[kCellsLds];
    __shared__ uint32_t psum[NT];
    __shared__ uint32_t scratch[NT / kWave + 1];
    __shared__ uint32_t s_T, s_over;
    const uint32_t h = blockIdx.x;
    if (h >= nlong || hmode[h] != kModeDirect) return;
    OSP_PLAN_DECL
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const uint64_t above = lanes_above_mask();
    const uint32_t i = rows[h];
    const uint32_t c0 = rowfirst[i], nc = rowfirst[i + 1] - c0;
    const uint32_t b = hbits[h], nfine = 1u << b, Ta = nseg[h];
    const int sh = colbits - (int)b;
    const uint32_t q0 = (uint32_t)hoff[h];   // the row's first record in the second buffer (the buffer holds < 2^32)
    uint32_t *__restrict__ rb = cells + cellbase[h];
    const uint32_t lutw = (nfine + 3) / 4;
    // descriptors of chunks [cb, cb + nb) -> cbs / cst (cst[nb] = entries of the block)
    auto load_block = [&](uint32_t cb, uint32_t nb) {
        uint32_t len[CBL / NT], bsv[CBL / NT], sum = 0;
#pragma unroll
        for (int q = 0; q < CBL / NT; q++) {
            const uint32_t j = tid * (CBL / NT) + q;   // blocked: a thread owns consecutive chunks
            len[q] = 0; bsv[q] = 0;
            if (j < nb) {
                const uint32_t c = c0 + cb + j;
                bsv[q] = ct_bs[c];
                len[q] = (uint32_t)(ct_off[c + 1] - ct_off[c]);
            }
            sum += len[q];
        }
        uint32_t total;
        uint32_t ex = block_excl_scan<uint32_t, NT>(sum, scratch, &total);
#pragma unroll
        for (int q = 0; q < CBL / NT; q++) {
            const uint32_t j = tid * (CBL / NT) + q;
            if (j < nb) { cst[j] = ex; cbs[j] = bsv[q] - ex; }   // cbs: entry number in the block -> position in B (one read per entry)
            ex += len[q];
        }
        if (tid == 0) cst[nb] = total;
        __syncthreads();
        return total;
    };
    // walks the block's entries; f(chunk in block, column, valid) is called with whole waves, valid lanes a prefix
    auto for_entries = [&](uint32_t nb, uint32_t E, auto f) {
        for (uint32_t base = w * (UNR * kWave); base < E; base += NT * UNR) {
            uint32_t cu[UNR], col[UNR];
            uint32_t c = direct_find_chunk(cst, nb, base);   // (wave-uniform)
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const uint32_t e = base + u * kWave + lane;
                const bool valid = e < E;
                while (valid && e >= cst[c + 1]) c++;
                cu[u] = c;
                col[u] = b_colidx[valid ? cbs[c] + e : 0u];   // clamped: the loads go out together
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) f(cu[u], col[u], base + u * kWave + lane < E);
        }
    };
    for (uint32_t d = tid; d <= nfine; d += NT) hist[d] = 0;
    __syncthreads();
    OSP_PLAN_MARK(0);   // row header
    // ---- 1. histogram over the fine bins
    uint32_t have_cb = 0, have_nb = 0, have_E = 0;   // the block of chunks whose descriptors are in LDS
    for (uint32_t cb = 0; cb < nc; cb += CBL) {
        const uint32_t nb = min((uint32_t)CBL, nc - cb);
        const uint32_t E = load_block(cb, nb);
        OSP_PLAN_MARK(1);   // chunk descriptors
        have_cb = cb; have_nb = nb; have_E = E;
        for_entries(nb, E, [&](uint32_t, uint32_t col, bool valid) {
#if OSP_PLAN_RUNS
            const uint32_t bin = valid ? col >> sh : 0xffffffffu;
            uint32_t runlen;
            if (wave_run_head(bin, valid, runlen, above)) atomicAdd(&hist[bin], runlen);
#else
            if (valid) atomicAdd(&hist[col >> sh], 1u);
#endif
        });
        __syncthreads();   // before the next block's descriptors replace these
        OSP_PLAN_MARK(2);   // histogram pass
    }
    // ---- 2. greedy grouping into ranges of at most `cap`: prefix sums, the bin every range starting at d ends before,
    // then one thread follows that chain (a few steps instead of one per bin)
    {
        constexpr int PER = NFINE / NT;   // bins per thread (blocked)
        static_assert(NFINE % NT == 0 && PER >= 1, "the fine bins are scanned PER per thread");
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) { const uint32_t d = tid * PER + q; v[q] = d < nfine ? hist[d] : 0u; sum += v[q]; }
        uint32_t total;
        uint32_t ex = block_excl_scan<uint32_t, NT>(sum, scratch, &total);
#pragma unroll
        for (int q = 0; q < PER; q++) { const uint32_t d = tid * PER + q; if (d < nfine) hist[d] = ex; ex += v[q]; }
        if (tid == 0) hist[nfine] = total;
    }
    __syncthreads();
    for (uint32_t d = tid; d < nfine; d += NT) {
        // the range that starts at bin d holds bins [d, e): the largest e with hist[e] - hist[d] <= cap, but at least one bin
        const uint32_t lim = hist[d] + cap;
        uint32_t lo = d + 1, hi = nfine + 1;   // first e in (d, nfine] with hist[e] > lim, or nfine + 1
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (hist[mid] <= lim) lo = mid + 1; else hi = mid; }
        nxt[d] = (uint16_t)max(d + 1, lo - 1);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0, over = 0;
        for (uint32_t d = 0; d < nfine; d = nxt[d]) { rbin0[t] = d; roff[t] = hist[d]; over |= (hist[nxt[d]] - hist[d]) > cap ? 1u : 0u; t++; }
        rbin0[t] = nfine; roff[t] = hist[nfine];
        s_T = t;
        s_over = over;   // a range that exceeds a tile (one bin does): its records must exist for the paths that take those
    }
    __syncthreads();
    const uint32_t T = s_T;   // <= Ta - 1 (split_params_kernel's bound)
    // (a range that exceeds a tile -- s_over -- is gathered like the others: expand_segments_kernel writes its records before the
    // paths for over-long segments read them; OSP_GATHER_OVER=0, a debugging aid, leaves such rows to the multiply as round 4 did)
    const bool gathered = gp.rdbase != nullptr && (s_over == 0 || gp.over);   // (workgroup-uniform)
    for (uint32_t d = tid; d < nfine; d += NT) {   // the range of every bin: last t with rbin0[t] <= d
        uint32_t lo = 0, hi = T;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rbin0[mid] <= d) lo = mid; else hi = mid; }
        lut[d] = (uint8_t)lo;
    }
    for (uint32_t t = tid; t < Ta; t += NT) {
        const uint64_t v = vbase[h] + t;
        const uint32_t tt = min(t, T);   // the unused segments at the end: empty
        vrow_off[v] = (uint64_t)q0 + roff[tt];
        vcol0[v] = (uint32_t)min((uint64_t)rbin0[tt] << sh, (uint64_t)0xffffffffu);
        vcol1[v] = (uint32_t)min((uint64_t)rbin0[min(tt + 1, T)] << sh, (uint64_t)0xffffffffu);
        if (t < T) cursor[t] = 0;
    }
    __syncthreads();
    if (gathered) {
        // ---- 3g. run descriptors (see above) ----
        constexpr uint32_t kAvMax = kBigCells ? (uint32_t)CBL : (uint32_t)((NFINE + 1) * sizeof(uint32_t) / (sizeof(V)));   // A values of a block of chunks: where the histogram was
        const uint32_t CB = max(1u, min(min((uint32_t)CBL, (uint32_t)kCellsLds / T), kAvMax));
        const bool multi = nc > CB;
        V *avs = kBigCells ? avs_own : reinterpret_cast<V *>(hist);
        uint32_t *nzc = reinterpret_cast<uint32_t *>(nxt), *ncur = rbin0;   // runs of every range (then: their exclusive prefix); runs written so far
        static_assert(sizeof(nxt) >= (kDirectMaxRanges + 1) * sizeof(uint32_t), "runs per range: where the grouping's chain was");
        const uint32_t rowbase = (uint32_t)gp.rdbase[h];
        // first column of every range (phase A's bisections; `cursor` is idle until phase B)
        static_assert(kDirectMaxRanges + 1 <= NT, "one range per thread");
        const uint32_t bcol_t = tid <= T ? (uint32_t)min((uint64_t)rbin0[tid] << sh, (uint64_t)0xffffffffu) : 0u;
        __syncthreads();
        if (tid <= T) { nzc[tid] = 0; ncur[tid] = 0; if (tid < T) cursor[tid] = bcol_t; }
        // (the A values of a row of ONE block of chunks are fetched now -- two dependent loads -- and arrive during the walk)
        if (!multi) { for (uint32_t cl = tid; cl < nc; cl += NT) avs[cl] = gp.av_in_order ? a_vals[c0 + cl] : a_vals[perm[c0 + cl]]; }
        __syncthreads();
        // A: (chunk, range) counts, block by block; runs per range
        for (uint32_t cb = 0; cb < nc; cb += CB) {
            const uint32_t nb = min(CB, nc - cb);
            for (uint32_t x = tid; x < nb * T; x += NT) cellm[x] = 0;
            uint32_t E = have_E;
            if (cb != have_cb || nb != have_nb) E = load_block(cb, nb);
            else __syncthreads();
            have_cb = cb; have_nb = nb; have_E = E;
            OSP_PLAN_MARK(1);
            // The block's (chunk, range) counts.  B's rows are sorted: a chunk's entries below a range's first column are a lower
            // bound, ~log2(length) dependent loads per (chunk, boundary) instead of a walk over every entry -- an eightieth of
            // the walk's instructions for a typical row (28 chunks of 177 entries, 3 ranges; the planner is bound by its vector
            // instructions), more than the walk only where the chunks are shorter than 8 entries per boundary: those walk.
            if (T > 1 && (uint64_t)E >= (uint64_t)nb * (T - 1) * 8u) {
                for (uint32_t x = tid; x < nb * (T - 1); x += NT) {
                    const uint32_t cl = x / (T - 1), t = x - cl * (T - 1) + 1;
                    const uint32_t col = cursor[t], b0 = cbs[cl] + cst[cl];
                    uint32_t lo = 0, hi = cst[cl + 1] - cst[cl];
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (b_colidx[b0 + mid] < col) lo = mid + 1; else hi = mid; }
                    lsm[cl * T + t] = lo;
                }
                for (uint32_t cl = tid; cl < nb; cl += NT) lsm[cl * T] = 0;
                __syncthreads();
                for (uint32_t x = tid; x < nb * T; x += NT) {
                    const uint32_t cl = x / T, t = x - cl * T;
                    cellm[x] = (t + 1 < T ? lsm[x + 1] : cst[cl + 1] - cst[cl]) - lsm[x];
                }
            } else {
                for_entries(nb, E, [&](uint32_t cl, uint32_t col, bool valid) {
                    const uint32_t key = valid ? __umul24(cl, T) + (uint32_t)lut[col >> sh] : 0xffffffffu;
                    uint32_t runlen;
                    if (wave_run_head(key, valid, runlen, above)) atomicAdd(&cellm[key], runlen);
                });
            }
            __syncthreads();
            OSP_PLAN_MARK(4);
            for (uint32_t x = tid; x < nb * T; x += NT) {
                const uint32_t c = cellm[x], cl = x / T, t = x - cl * T;
                if (c) atomicAdd(&nzc[t], 1u);
                if (multi) rb[lutw + (uint64_t)(cb + cl) * Ta + t] = c;
            }
            __syncthreads();
        }
        {
            const uint32_t v = tid < T ? nzc[tid] : 0u;
            uint32_t total;
            const uint32_t ex = block_excl_scan<uint32_t, NT>(v, scratch, &total);
            if (tid < T) nzc[tid] = ex;
            if (tid == 0) {
                nzc[T] = total;
                gp.rowruns[h] = total;
            }
            if (tid < T) cursor[tid] = 0;   // (the ranges' first columns have been used)
        }
        __syncthreads();
        for (uint32_t t = tid; t < Ta; t += NT) {
            const uint64_t v = vbase[h] + t;
            gp.vrun_off[v] = rowbase + nzc[min(t, T)];
            gp.vrun_end[v] = rowbase + nzc[min(t + 1, T)];
        }
        // B: prefixes and descriptors, block by block
        const uint32_t G = max(1u, min((uint32_t)NT / T, 16u));
        for (uint32_t cb = 0; cb < nc; cb += CB) {
            const uint32_t nb = min(CB, nc - cb);
            if (multi) {
                if (cb != have_cb || nb != have_nb) { have_E = load_block(cb, nb); have_cb = cb; have_nb = nb; }
                for (uint32_t x = tid; x < nb * T; x += NT) { const uint32_t cl = x / T, t = x - cl * T; cellm[x] = rb[lutw + (uint64_t)(cb + cl) * Ta + t]; }
            }
            if (multi) { for (uint32_t cl = tid; cl < nb; cl += NT) avs[cl] = gp.av_in_order ? a_vals[c0 + cb + cl] : a_vals[perm[c0 + cb + cl]]; }
            __syncthreads();
            for (uint32_t cl = tid; cl < nb; cl += NT) {   // where every range starts inside its chunk
                uint32_t run = 0;
                for (uint32_t t = 0; t < T; t++) { lsm[cl * T + t] = run; run += cellm[cl * T + t]; }
            }
            const uint32_t S = (nb + G - 1) / G;
            for (uint32_t t0 = 0; t0 < T; t0 += NT) {
                const uint32_t Tb = min((uint32_t)NT, T - t0);
                const uint32_t g = tid / Tb, t = t0 + (tid - g * Tb);
                const bool on = g < G;
                uint32_t run0 = 0, nrun0 = 0;
                __syncthreads();
                if (on) {
                    uint32_t sum = 0, nz = 0;   // (a row holds at most 2^20 products -- kDirectDenseMax --, a block at most CBL chunks: 21 + 11 bits)
                    static_assert(kDirectDenseMax <= (1ull << 20) && CBL < 2048, "products and runs of a group of chunks in one word");
                    for (uint32_t cl = g * S; cl < min(nb, (g + 1) * S); cl++) { const uint32_t c = cellm[cl * T + t]; sum += c; nz += c != 0; }
                    psum[tid] = sum | (nz << 21);
                    run0 = cursor[t];
                    nrun0 = ncur[t];
                }
                __syncthreads();
                if (on) {
                    uint32_t run = run0, nrun = nrun0;
                    for (uint32_t gg = 0; gg < g; gg++) { const uint32_t p = psum[gg * Tb + (t - t0)]; run += p & 0x1fffffu; nrun += p >> 21; }
                    const uint32_t rbase = q0 + roff[t];
                    RunDesc<V> *__restrict__ out = runs + rowbase + nzc[t];
                    for (uint32_t cl = g * S; cl < min(nb, (g + 1) * S); cl++) {
                        const uint32_t c = cellm[cl * T + t];
                        if (c) {
                            RunDesc<V> rd;
                            rd.dst = rbase + run;
                            rd.src = cbs[cl] + cst[cl] + lsm[cl * T + t];
                            rd.av = avs[cl];
                            out[nrun++] = rd;
                        }
                        run += c;
                    }
                    if (g == G - 1) { cursor[t] = run; ncur[t] = nrun; }
                }
            }
            if (gp.mark_skipped) { for (uint32_t cl = tid; cl < nb; cl += NT) chunk_off[perm[c0 + cb + cl]] = kChunkSkip; }
            __syncthreads();
            OSP_PLAN_MARK(5);
        }
        return;
    }
    for (uint32_t d = tid; d < lutw * 4; d += NT) reinterpret_cast<uint8_t *>(rb)[d] = d < nfine ? lut[d] : (uint8_t)0;
    if (tid == 0 && gp.rdbase != nullptr && gp.nwritten) atomicAdd(gp.nwritten, 1u);   // (rare: the panel's other direct rows are gathered)
    OSP_PLAN_MARK(3);   // grouping, segment tables
    // ---- 3. cells, a block of chunks at a time
    const uint32_t CB = max(1u, min((uint32_t)CBL, (uint32_t)kCellsLds / T));
    // column prefixes by (group of chunks, range): G groups of S chunks each
    const uint32_t G = max(1u, min((uint32_t)NT / T, 16u));   // (T > NT: one group, the ranges in rounds of NT)
    for (uint32_t cb = 0; cb < nc; cb += CB) {
        const uint32_t nb = min(CB, nc - cb);
        for (uint32_t x = tid; x < nb * T; x += NT) cellm[x] = 0;
        uint32_t E = have_E;
        if (cb != have_cb || nb != have_nb) E = load_block(cb, nb);   // (most rows: one block, still there from the histogram)
        else __syncthreads();
        have_cb = cb; have_nb = nb; have_E = E;
        OSP_PLAN_MARK(1);
        for_entries(nb, E, [&](uint32_t cl, uint32_t col, bool valid) {
#if OSP_PLAN_RUNS
            const uint32_t key = valid ? __umul24(cl, T) + (uint32_t)lut[col >> sh] : 0xffffffffu;   // the (chunk, range) cell; both factors < 2^24
            uint32_t runlen;
            if (wave_run_head(key, valid, runlen, above)) atomicAdd(&cellm[key], runlen);
#else
            if (valid) atomicAdd(&cellm[__umul24(cl, T) + (uint32_t)lut[col >> sh]], 1u);   // (chunk, range): both below 2^24
#endif
        });
        __syncthreads();
        OSP_PLAN_MARK(4);   // cell pass
        // where every range starts inside its chunk
        for (uint32_t cl = tid; cl < nb; cl += NT) {
            uint32_t run = 0;
            for (uint32_t t = 0; t < T; t++) { lsm[cl * T + t] = run; run += cellm[cl * T + t]; }
        }
        // where every chunk's run starts inside its range: sums per group of chunks, then the walk (NT ranges at a time)
        const uint32_t S = (nb + G - 1) / G;
        for (uint32_t t0 = 0; t0 < T; t0 += NT) {
            const uint32_t Tb = min((uint32_t)NT, T - t0);   // ranges of this round; G groups fit when T <= NT (else G = 1)
            const uint32_t g = tid / Tb, t = t0 + (tid - g * Tb);
            const bool on = g < G;
            uint32_t run0 = 0;
            __syncthreads();   // psum of the previous round has been read
            if (on) {
                uint32_t sum = 0;
                for (uint32_t cl = g * S; cl < min(nb, (g + 1) * S); cl++) sum += cellm[cl * T + t];
                psum[tid] = sum;
                run0 = cursor[t];   // read before the barrier: the last group rewrites it behind it
            }
            __syncthreads();
            if (on) {
                uint32_t run = run0;
                for (uint32_t gg = 0; gg < g; gg++) run += psum[gg * Tb + (t - t0)];
                const uint32_t rbase = q0 + roff[t];
                for (uint32_t cl = g * S; cl < min(nb, (g + 1) * S); cl++) {
                    rb[lutw + (uint64_t)(cb + cl) * Ta + t] = rbase + run - lsm[cl * T + t];
                    run += cellm[cl * T + t];
                }
                if (g == G - 1) cursor[t] = run;   // (the last group ends at the block's end, possibly with no chunk of its own)
            }
        }
        for (uint32_t cl = tid; cl < nb; cl += NT)
            chunk_off[perm[c0 + cb + cl]] = direct_desc((uint32_t)cellbase[h], lutw + (cb + cl) * Ta, (uint32_t)sh);
        __syncthreads();   // the block's cells and descriptors have been read
        OSP_PLAN_MARK(5);   // prefixes, cells and descriptors out
    }
}
// ---- hub rows: the plan the multiply phase writes them by -----------------------------------------------------------------
// (what a hub row is and what the multiply does with the cells: osp_kernels.h, "HUB rows")
// The run table of B for blocks of 2^sh columns, made once per product: flag(e) = entry e starts a run (first entry of its row,
// or another block than the entry before it); sx = exclusive scan of the flags; runstart[g] = first entry of run g.
__global__ void hub_rowstart_kernel(const int64_t *b_rowptr, uint64_t K, uint8_t *rowstart) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K && b_rowptr[k + 1] > b_rowptr[k]) rowstart[b_rowptr[k]] = 1;
}
struct HubRunFlag {
    const uint8_t *rowstart;
    const uint32_t *col;
    int sh;
    __device__ uint32_t operator()(uint64_t e) const { return (rowstart[e] || (col[e] >> sh) != (col[e - 1] >> sh)) ? 1u : 0u; }   // (rowstart[0] is set)
};
__global__ void hub_runstart_end_kernel(const uint32_t *nruns_p, uint32_t *runstart, uint32_t nnz) { runstart[*nruns_p] = nnz; }
// the column block of every run (at most 2^kSplitMaxBits blocks): the planner's walks read it beside the run's bounds instead of
// following the run's first entry into B's column indices -- a dependent, scattered 4-byte load per run
__global__ void hub_runblk_kernel(const uint32_t *__restrict__ nruns_p, const uint32_t *__restrict__ runstart, const uint32_t *__restrict__ b_colidx,
                                  int sh, uint16_t *__restrict__ runblk) {
    const uint32_t nruns = *nruns_p;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nruns; g += gridDim.x * blockDim.x) runblk[g] = (uint16_t)(b_colidx[runstart[g]] >> sh);
}
struct HubTables {
    const uint32_t *sx = nullptr;         // runs that start before entry e: nnz(B) + 1 entries
    const uint32_t *runstart = nullptr;   // first entry of every run, and nnz(B) behind the last
    const uint16_t *runblk = nullptr;     // column block of every run (one more entry behind the last: never used, readable)
    int sh = 0;                           // block width: 2^sh columns
};
// One WAVE per job of a hub row (the stretch rows' jobs: job st of a row = the chunks whose records start inside the row's
// products [st, st + 1) * kSplitJob).  The job's runs -- chunk after chunk in ascending k, a chunk's runs in column order: the
// order the records of a block must have -- are walked 64 at a time: lane q holds chunk q's first run and run count, a lane
// finds the chunk of flat run x by bisecting the prefix of the counts.
//   CELLS = false  per-block totals of the job -> ghist[hbase + block * nst + st] (the stretch split's histogram layout, so the
//                  segment offsets come out of the same scan and split_vrows_kernel), and the job's number of runs;
//   CELLS = true   ghist holds the scanned offsets: every run's place is what ONE LDS atomic on its block's running offset
//                  returns -- lanes hitting the same block get their old values in ascending lane order (the property the radix
//                  ranks rest on, self-tested per context; without it the panel keeps the stretch split) -- and goes, minus
//                  the index of the run's first entry, into the chunk's cells; the chunk's descriptor goes to chunk_off.
constexpr int kHubThreads = kWave;
template <bool CELLS>
__global__ __launch_bounds__(kHubThreads) void hub_plan_kernel(
    const uint32_t *__restrict__ rows, uint32_t nheavy, const uint64_t *__restrict__ blkbase, const uint64_t *__restrict__ hbase,
    const uint8_t *__restrict__ hbits, const uint32_t *__restrict__ nstretch, const uint64_t *__restrict__ row_off,
    const uint32_t *__restrict__ rowfirst, const uint64_t *__restrict__ ct_off, const uint32_t *__restrict__ ct_bs,
    const uint32_t *__restrict__ perm, const uint32_t *__restrict__ b_colidx, const HubTables ht, uint32_t *__restrict__ ghist,
    const uint64_t *__restrict__ hoff, uint64_t *__restrict__ jobruns, const uint64_t *__restrict__ jobcell, uint32_t *__restrict__ cells,
    uint64_t *__restrict__ chunk_off) {
    extern __shared__ uint32_t hist[];   // 2^hub_b running offsets (dynamic: 8 KB for 2048 blocks -- 20 jobs in flight per CU)
    const unsigned lane = lane_id();
    const uint32_t h = (uint32_t)(upper_bound_dev(blkbase, 0, (uint64_t)nheavy + 1, (uint64_t)blockIdx.x) - 1);
    const uint32_t st = (uint32_t)(blockIdx.x - blkbase[h]), nst = nstretch[h];
    const uint32_t nseg = 1u << hbits[h];
    const uint32_t row = rows[h];
    const uint64_t hb = hbase[h];
    // the job's chunks: those whose records start in [jbeg, jend) (absolute staging offsets; the last job takes the rest)
    const uint64_t jbeg = row_off[row] + (uint64_t)st * kSplitJob, jend = st + 1 == nst ? ~0ull : jbeg + (uint64_t)kSplitJob;
    const uint32_t c0 = rowfirst[row], c1 = rowfirst[row + 1];
    const uint32_t ta = (uint32_t)lower_bound_dev(ct_off, (uint64_t)c0, (uint64_t)c1, jbeg);
    const uint32_t tb = jend == ~0ull ? c1 : (uint32_t)lower_bound_dev(ct_off, (uint64_t)ta, (uint64_t)c1, jend);
    if (CELLS) {
        const uint32_t qrow = (uint32_t)hoff[h] - ghist[hb];
        for (uint32_t d = lane; d < nseg; d += kWave) hist[d] = qrow + ghist[hb + (uint64_t)d * nst + st];
    } else {
        for (uint32_t d = lane; d < nseg; d += kWave) hist[d] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    uint64_t done = 0;   // runs of the job's earlier chunk batches
    const uint64_t cell0 = CELLS ? jobcell[blockIdx.x] : 0ull;
    // The walk is a chain of dependent loads per batch of 64 chunks (chunk table -> run table -> first entries -> columns), and
    // a job of one-entry chunks has hundreds of batches: the chunk-table loads of batch t + 2 and the run-table loads of batch
    // t + 1 are in flight while batch t's runs are walked, four groups of 64 runs at a time.
    constexpr int XU = 4;
    auto load1 = [&](uint32_t t0, uint32_t &bs, uint32_t &nb) {   // chunk table: B row and length of chunk t0 + lane
        const uint32_t t = t0 + lane;
        bs = 0; nb = 0;
        if (t0 < tb && t < tb) { bs = ct_bs[t]; nb = (uint32_t)(ct_off[t + 1] - ct_off[t]); }
    };
    auto load2 = [&](uint32_t bs, uint32_t nb, uint32_t &g0, uint32_t &g1) {   // run table: the chunk's first run and the one behind its last
        g0 = ht.sx[bs];                       // (clamped, branch-free: a chunk without entries reads two equal values... of entry bs)
        g1 = ht.sx[(uint64_t)bs + nb];
    };
    uint32_t bs_a, nb_a, bs_b, nb_b, g0_a, g1_a;
    load1(ta, bs_a, nb_a);
    load1(ta + kWave, bs_b, nb_b);
    load2(bs_a, nb_a, g0_a, g1_a);
    for (uint32_t t0 = ta; t0 < tb; t0 += kWave) {
        const uint32_t t = t0 + lane;
        const uint32_t bs = bs_a, g0 = g0_a, nr = g1_a - g0_a;   // (a chunk without entries: no runs)
        // the next batch's run table and the chunk table of the batch after it go out now
        uint32_t g0_n, g1_n, bs_c, nb_c;
        load2(bs_b, nb_b, g0_n, g1_n);
        load1(t0 + 2 * kWave, bs_c, nb_c);
        const uint32_t incl = wave_incl_scan(nr), pre = incl - nr;
        const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (CELLS && t < tb) chunk_off[perm[t]] = hub_desc(cell0 + done + pre);
        for (uint32_t x0 = 0; x0 < T; x0 += XU * kWave) {
            uint32_t e0[XU], e1[XU], cbs[XU], blk[XU];
#pragma unroll
            for (int u = 0; u < XU; u++) {
                const uint32_t x = x0 + u * kWave + lane;
                uint32_t lo = 0, hi = kWave;   // last lane whose runs start at or before x (lanes without runs never win)
#pragma unroll
                for (int step = 0; step < 6; step++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint32_t)__shfl((int)pre, (int)mid) <= x) lo = mid; else hi = mid;
                }
                const uint32_t xr = x - (uint32_t)__shfl((int)pre, (int)lo);
                const uint32_t gl = (uint32_t)__shfl((int)g0, (int)lo);   // (every lane takes part in the shuffle: never under `x < T`)
                const uint32_t g = x < T ? gl + xr : 0u;                  // clamped: the loads go out together
                cbs[u] = (uint32_t)__shfl((int)bs, (int)lo);
                e0[u] = ht.runstart[g];
                e1[u] = ht.runstart[g + 1];
                blk[u] = ht.runblk[g];
            }
#pragma unroll
            for (int u = 0; u < XU; u++) {
                const uint32_t x = x0 + u * kWave + lane;
                if (x < T) {
                    const uint32_t old = atomicAdd(&hist[blk[u]], e1[u] - e0[u]);   // (lanes and groups in flat run order: the order inside a block)
                    if (CELLS) cells[cell0 + done + x] = old - (e0[u] - cbs[u]);
                }
            }
        }
        done += T;
        bs_a = bs_b; nb_a = nb_b; g0_a = g0_n; g1_a = g1_n;
        bs_b = bs_c; nb_b = nb_c;
    }
    __builtin_amdgcn_wave_barrier();
    if (!CELLS) {
        for (uint32_t d = lane; d < nseg; d += kWave) ghist[hb + (uint64_t)d * nst + st] = hist[d];
        if (lane == 0) jobruns[blockIdx.x] = done;
    }
}
}  // namespace osp
