// osp_api.hip -- C ABI (include/outerspace_spgemm.h) and host orchestration of the GPU pipeline.
//
// Reference call stack being replaced:
//   parts = cscMulcsr(csc, csr)      SimSpGEMM.cpp:265-281   -> symbolic + multiply_kernel
//   C     = deduplicateCOO(concat)   SimSpGEMM.cpp:519-535   -> merge_tiles_kernel / global sort
// There is no CPU fallback here: every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/outerspace_spgemm.h"
#include "osp_internal.h"
#include "osp_kernels.h"
#include "osp_split.h"
#include "osp_sort.h"
#include "osp_epilogue.h"

namespace osp {

// (fail() and the per-thread error string live in osp_host.cpp, the host-only TU)

#define OSP_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            throw Error(e_ == hipErrorOutOfMemory ? OSP_ERR_ALLOC : OSP_ERR_HIP,           \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

#include "osp_context.h"
#include "osp_pipeline.h"

// Copies an input array to the device when it lives on the host.
template <class T>
static const T *to_device(Scratch &sc, const T *p, uint64_t n, osp_memspace_t space, hipStream_t s) {
    if (space == OSP_DEVICE || n == 0) return p;
    T *d = sc.get<T>(n);
    copy_h2d(d, p, n * sizeof(T), s);
    return d;
}

static void check_flags(uint32_t f, const char *what) {
    if (f & kFlagPtr) throw Error(OSP_ERR_ARG, std::string(what) + ": pointer array is not a monotone 0..nnz sequence");
    if (f & kFlagRange) throw Error(OSP_ERR_RANGE, std::string(what) + ": index outside its dimension");
    if (f & kFlagDuplicate) throw Error(OSP_ERR_DUPLICATE, std::string(what) + ": duplicate coordinate (reference: throw(233))");
    if (f & kFlagUnsorted) throw Error(OSP_ERR_UNSORTED, std::string(what) + ": indices inside a segment are not ascending");
}

// *out += sum_k nnz(A[:,k]) * nnz(B[k,:]) over all k (one atomic per workgroup)
__global__ void count_partials_kernel(const int64_t *a_colptr, const int64_t *b_rowptr, uint64_t K, unsigned long long *out) {
    __shared__ uint64_t scratch[256 / kWave + 1];
    uint64_t sum = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += (uint64_t)gridDim.x * blockDim.x)
        sum += (uint64_t)(a_colptr[k + 1] - a_colptr[k]) * (uint64_t)(b_rowptr[k + 1] - b_rowptr[k]);
    uint64_t total;
    block_excl_scan<uint64_t, 256>(sum, scratch, &total);
    if (threadIdx.x == 0 && total) atomicAdd(out, (unsigned long long)total);
}

template <class T>
static void spgemm_impl(Context *ctx, Result *res, uint64_t M, uint64_t K, uint64_t N, const int64_t *a_colptr_in,
                        const uint32_t *a_rowidx_in, const T *a_vals_in, const int64_t *b_rowptr_in,
                        const uint32_t *b_colidx_in, const T *b_vals_in, osp_memspace_t space,
                        const osp_config_t &cfg, const PanelSink *sink = nullptr, bool partials_only = false) {
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    PhaseTimer tm(s);
    EventPair ev;
    OSP_HIP(hipEventRecord(ev.a, s));

    // pointer arrays first: nnz comes from their last entries
    const int64_t *a_colptr = to_device(sc, a_colptr_in, K + 1, space, s);
    const int64_t *b_rowptr = to_device(sc, b_rowptr_in, K + 1, space, s);
    int64_t nnz_a, nnz_b;
    // (with them comes the number of partial products of the whole product -- a sum over k of two differences: what decides
    // whether a product of few non-zeros still plans direct rows, see `direct` below)
    uint64_t p_all = 0;
    unsigned long long *p_all_dev = (unsigned long long *)sc.get<uint64_t>(1);
    zero_async(s, {{p_all_dev, sizeof(uint64_t)}});
    // (few workgroups: every one ends in an atomic on ONE word, 11-13 ns apiece -- 3 579 of them were the 46 us this kernel took
    // on the web-Google shape)
    if (K) count_partials_kernel<<<(unsigned)std::min<uint64_t>(grid_for(K, 256), 256), 256, 0, s>>>(a_colptr, b_rowptr, K, p_all_dev);
    {
        Gather g(s);
        if (space == OSP_HOST) { nnz_a = a_colptr_in[K]; nnz_b = b_rowptr_in[K]; }
        else { g.add(&nnz_a, a_colptr + K); g.add(&nnz_b, b_rowptr + K); }
        g.add(&p_all, (const uint64_t *)p_all_dev);
        g.wait();
    }
    if (nnz_a < 0 || nnz_b < 0) throw Error(OSP_ERR_ARG, "negative nnz in pointer array");
    if ((uint64_t)nnz_a >= 0xffffffffull || (uint64_t)nnz_b >= 0xffffffffull)
        throw Error(OSP_ERR_ARG, "operands with >= 2^32 non-zeros are not supported");
    const uint32_t *a_rowidx = to_device(sc, a_rowidx_in, nnz_a, space, s);
    const T *a_vals = to_device(sc, a_vals_in, nnz_a, space, s);
    const uint32_t *b_colidx = to_device(sc, b_colidx_in, nnz_b, space, s);
    const T *b_vals = to_device(sc, b_vals_in, nnz_b, space, s);
    res->info.nnz_a = nnz_a;
    res->info.nnz_b = nnz_b;

    if (cfg.validate) {
        uint32_t *flags = sc.get<uint32_t>(2);
        OSP_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(uint32_t), s));
        validate_ptr_kernel<<<grid_for(K + 1, 256), 256, 0, s>>>(a_colptr, K, nnz_a, flags);
        validate_ptr_kernel<<<grid_for(K + 1, 256), 256, 0, s>>>(b_rowptr, K, nnz_b, flags + 1);
        uint32_t fa = 0, fb = 0;
        { Gather g(s); g.add(&fa, (const uint32_t *)flags); g.add(&fb, (const uint32_t *)flags + 1); g.wait(); }
        check_flags(fa, "A (CSC)");
        check_flags(fb, "B (CSR)");
        if (nnz_a) validate_idx_kernel<<<grid_for(nnz_a, 256), 256, 0, s>>>(a_colptr, a_rowidx, K, nnz_a, M, flags);
        if (nnz_b) validate_idx_kernel<<<grid_for(nnz_b, 256), 256, 0, s>>>(b_rowptr, b_colidx, K, nnz_b, N, flags + 1);
        { Gather g(s); g.add(&fa, (const uint32_t *)flags); g.add(&fb, (const uint32_t *)flags + 1); g.wait(); }
        check_flags(fa, "A (CSC)");
        check_flags(fb, "B (CSR)");
    }

    // ---- k shard ----
    const uint64_t k0 = cfg.k_begin, k1 = cfg.k_end ? cfg.k_end : K;
    if (getenv("OSP_VERBOSE"))
        fprintf(stderr, "[osp] spgemm M=%llu K=%llu N=%llu nnzA=%lld nnzB=%lld k=[%llu,%llu) %s operands\n", (unsigned long long)M,
                (unsigned long long)K, (unsigned long long)N, (long long)nnz_a, (long long)nnz_b, (unsigned long long)k0,
                (unsigned long long)k1, space == OSP_HOST ? "host" : "device");
    if (k0 > k1 || k1 > K) throw Error(OSP_ERR_ARG, "k range outside [0,K]");
    int64_t e0 = 0, e1 = nnz_a;
    if (k0 != 0 || k1 != K) {
        if (space == OSP_HOST) { e0 = a_colptr_in[k0]; e1 = a_colptr_in[k1]; }
        else { Gather g(s); g.add(&e0, a_colptr + k0); g.add(&e1, a_colptr + k1); g.wait(); }
    }
    // ---- row-sharded multi-GPU mode: find this rank's rows and drop the rest of A BEFORE the symbolic phase ----
    // (a rank then sorts 1/G of A's non-zeros instead of all of them; every rank derives the same bounds from the
    // replicated operands alone: no collective)
    uint64_t r_lo = 0, r_hi = M;
    const bool row_sharded = cfg.row_shard_count > 1;
    if (row_sharded) {
        const uint64_t nnz = (uint64_t)(e1 - e0);  // all of the k shard's non-zeros: the pre-pass sees every row
        if (cfg.row_shard_index < 0 || cfg.row_shard_index >= cfg.row_shard_count) throw Error(OSP_ERR_ARG, "row shard index out of range");
        const uint32_t G = (uint32_t)cfg.row_shard_count;
        tm.begin(PH_SYM);
        {
            Scratch cs(ctx);
            unsigned long long *work = (unsigned long long *)cs.get<uint64_t>(M + 1);
            uint64_t *pre = cs.get<uint64_t>(M + 1), *cost_pre = cs.get<uint64_t>(M + 1);
            uint64_t *tmp = cs.get<uint64_t>(scan_scratch_entries(M + 1));
            uint64_t *d_b = cs.get<uint64_t>(2ull * (G + 1));
            OSP_HIP(hipMemsetAsync(work, 0, (M + 1) * sizeof(uint64_t), s));
            // every 16th column is sample enough to balance G shards of a large matrix; small ones are counted exactly
            const uint32_t stride = (k1 - k0) >= (1u << 16) ? 16u : 1u;
            const uint64_t nsample = (k1 - k0 + stride - 1) / stride;
            if (nnz) row_work_kernel<<<grid_for(nsample * kWave, 256), 256, 0, s>>>(a_colptr, a_rowidx, b_rowptr, k0, k1, stride, work);
            device_exclusive_scan<LoadU64, uint64_t>(LoadU64{(const uint64_t *)work}, M, pre, tmp, s);
            device_exclusive_scan<RowCost, uint64_t>(RowCost{pre, (uint64_t)TileCap<T>::value, kSplitRowMax}, M, cost_pre, tmp, s);
            shard_bounds_kernel<<<grid_for(G + 1, 64), 64, 0, s>>>(cost_pre, pre, M, G, d_b, d_b + G + 1);
            std::vector<uint64_t> h_b(2ull * (G + 1));
            copy_d2h(h_b.data(), d_b, h_b.size() * sizeof(uint64_t), s);
            r_lo = h_b[cfg.row_shard_index];
            r_hi = h_b[cfg.row_shard_index + 1];
        }
        // A restricted to rows [r_lo, r_hi): same K columns, absolute row ids
        int64_t *colptr2 = sc.get<int64_t>(K + 1);
        uint64_t nnz2 = 0;
        uint32_t *rowidx2 = nullptr;
        T *vals2 = nullptr;
        {
            Scratch cs(ctx);
            uint32_t *keep_scan = cs.get<uint32_t>(nnz + 1);
            uint32_t *tmp = cs.get<uint32_t>(scan_scratch_entries(nnz + 1));
            const RowInRange keep{a_rowidx + e0, (uint32_t)r_lo, r_hi};
            device_exclusive_scan<RowInRange, uint32_t>(keep, nnz, keep_scan, tmp, s);
            nnz2 = d2h(keep_scan + nnz, s);
            rowidx2 = sc.get<uint32_t>(nnz2);
            vals2 = sc.get<T>(nnz2);
            restrict_colptr_kernel<<<grid_for(K + 1, 256), 256, 0, s>>>(a_colptr, K, e0, nnz, keep_scan, colptr2);
            if (nnz2) restrict_compact_kernel<T><<<grid_for(nnz, 256), 256, 0, s>>>(keep, keep_scan, nnz, a_vals + e0, rowidx2, vals2);
        }
        tm.end(PH_SYM);
        a_colptr = colptr2; a_rowidx = rowidx2; a_vals = vals2;
        e0 = 0;  // columns before k0 are empty now
        e1 = (int64_t)nnz2;
    }
    const uint64_t nnz = (uint64_t)(e1 - e0);  // non-zeros of A inside the shard

    // ---- symbolic: chunk offsets in (row, k) order ----
    tm.begin(PH_SYM);
    uint64_t *row_off = sc.get<uint64_t>(M + 1);
    uint64_t *chunk_off = sc.get<uint64_t>(nnz);
    uint64_t P = 0;
    // Row-wise variant (cfg.algorithm): rows of up to one tile of partial products are computed inside the tile kernel
    // from the chunk table; B's offsets must fit 32 bits for it (otherwise the outer-product path runs as usual)
    const int algo = cfg.algorithm;
    if (algo != OSP_ALGO_OUTER && algo != OSP_ALGO_ROWWISE) throw Error(OSP_ERR_ARG, "unknown algorithm");
    const bool rowwise = algo == OSP_ALGO_ROWWISE && nnz && (uint64_t)nnz_b < 0xffffffffull && nnz < 0xffffffffull && !partials_only;
    ChunkTable<T> ct{};
    // Long rows that one workgroup could split are written straight into their column ranges by the multiply phase
    // ("direct" rows, osp_split.h) when the operands allow 32-bit B offsets.  OSP_DIRECT=0 switches that off (every long
    // row is then split after the multiply, as the parts-merging entry points do), OSP_DIRECT_MAX=<partial products>
    // bounds the rows it applies to.
    // Small products keep the split: the plan costs a handful of launches and a read-back, which a product of a few
    // milliseconds does not earn back (web-Google shape: 2.3 ms with the split, 2.7 with direct rows).  OSP_DIRECT_MIN_NNZ
    // moves that boundary (the tests set it to 0, so that their small inputs take the direct path).
    const uint64_t direct_min_nnz = getenv("OSP_DIRECT_MIN_NNZ") ? strtoull(getenv("OSP_DIRECT_MIN_NNZ"), nullptr, 10) : ((getenv("OSP_GATHER") && atoi(getenv("OSP_GATHER")) == 0) ? (8ull << 20) : (2ull << 20));
    // ... unless its output rows are dense on average (at least 0.375 partial products per entry of the M x N result -- three times the
    // density from which a long row's column ranges are capped at the dense accumulators' width, plan_panel): such rows are
    // written in a few wide ranges, long runs, and summed without a sort -- 4096^2 with 880 entries per row (3.6 M non-zeros,
    // 3.2 G partial products) 44.4 -> 26.4 ms, Graph500 scale 14 ef 512 100 -> 79 ms.
    const bool dense_avg = (long double)p_all * 8.0L >= 3.0L * (long double)M * (long double)N;
    const bool direct = nnz && (nnz >= direct_min_nnz || dense_avg) && (uint64_t)nnz_b < 0xffffffffull && nnz < 0xffffffffull && !partials_only &&
                        !(getenv("OSP_DIRECT") && atoi(getenv("OSP_DIRECT")) == 0);
    const uint64_t direct_max = std::min<uint64_t>(getenv("OSP_DIRECT_MAX") ? strtoull(getenv("OSP_DIRECT_MAX"), nullptr, 10) : kSplitRowMax,
                                                   kDirectDenseMax);   // (the planner counts a row's products in 21 bits)
    // Gathered rows (osp_kernels.h): the merge kernel forms the partial products of planned long rows and of short rows itself,
    // from run descriptors; on unless OSP_GATHER=0 (debugging aid, A/B timing: every row is then written by the multiply, as
    // until round 4) or the row-wise variant runs (its tile kernel is another instantiation).  OSP_GATHER=1: long rows only.
    const int gather_env = getenv("OSP_GATHER") ? atoi(getenv("OSP_GATHER")) : 2;
    const bool gather_ok = nnz && !rowwise && !partials_only && (uint64_t)nnz_b < 0xffffffffull && nnz < 0xffffffffull;
    // (short rows: only where long rows are planned too -- a product of a few million non-zeros does not earn the tables back:
    // web-Google shape 2.09 -> 2.50 ms with them)
    const bool gather_short = gather_ok && gather_env >= 2 && direct;
    ShortRuns<T> srun{};
    DirectSrc dsrc{};
    uint32_t n_long_rows = 1;
    if (nnz == 0) {
        OSP_HIP(hipMemsetAsync(row_off, 0, (M + 1) * sizeof(uint64_t), s));
    } else {
        Scratch ss(ctx);
        Scratch &keep = (rowwise || direct || gather_short) ? sc : ss;  // the chunk table outlives the symbolic phase
        uint32_t *ka = ss.get<uint32_t>(nnz), *pa = ss.get<uint32_t>(nnz), *kb = ss.get<uint32_t>(nnz), *pb = ss.get<uint32_t>(nnz);
        uint32_t *rows_sorted = (gather_short ? keep : ss).template get<uint32_t>(nnz), *perm = keep.get<uint32_t>(nnz), *w_sorted = ss.get<uint32_t>(nnz);
        uint32_t *bs_sorted = (rowwise || direct || gather_short) ? keep.get<uint32_t>(nnz) : nullptr;
        uint32_t *rowfirst = keep.get<uint32_t>(M + 1);
        uint32_t *hist = ss.get<uint32_t>(rs_hist_entries(nnz));
        uint32_t *hist_tmp = ss.get<uint32_t>(scan_scratch_entries(rs_hist_entries(nnz)));
        uint64_t *offs_sorted = keep.get<uint64_t>(nnz + 1);
        uint64_t *scan_tmp = ss.get<uint64_t>(scan_scratch_entries(std::max<uint64_t>(nnz, M + 1)));
        // (row, k) order of A's non-zeros; the last sort pass also looks up each chunk's length
        const bool table = rowwise || direct || gather_short;   // keep the chunk table: (length, B row) pairs in `w`
        // (gathered rows: the A values ride along, so that the run descriptors' makers read them in (row, k) order)
        const bool with_av = gather_ok && gather_env >= 1 && (direct || gather_short);
        T *av_sorted = with_av ? keep.get<T>(nnz) : nullptr;
        uint32_t *w = ss.get<uint32_t>(with_av ? 4 * nnz : table ? 2 * nnz : nnz);
        uint32_t *bs = table ? w : nullptr;
        sym_chunk_len_kernel<<<grid_for(nnz, 256), 256, 0, s>>>(a_colptr, b_rowptr, k0, k1, e0, nnz, w, bs,
                                                                with_av ? reinterpret_cast<const uint32_t *>(a_vals) : nullptr, (uint32_t)(sizeof(T) / 4));
        SymEpilogue sym_ep{w, bs, rows_sorted, perm, w_sorted, bs_sorted};
        if (with_av) { sym_ep.av_sorted = av_sorted; sym_ep.vwords = (uint32_t)(sizeof(T) / 4); }
        device_sort_rows<SymEpilogue>(a_rowidx + e0, nnz, std::max(1, bits_for(M)), ka, pa, kb, pb, hist, hist_tmp, sym_ep, s, ctx->rank_atomic);
        device_exclusive_scan<LoadU32As64, uint64_t>(LoadU32As64{w_sorted}, nnz, offs_sorted, scan_tmp, s);
        sym_row_offsets_kernel<<<grid_for(M + 1, 256), 256, 0, s>>>(rows_sorted, offs_sorted, nnz, M, row_off, rowfirst);
        const uint64_t rw_cap = (rowwise || gather_short) ? (uint64_t)TileCap<T>::value : 0ull;   // rows the multiply skips
        // (lazily where only rows written through cells would read them: DirectSrc::ensure_chunk_off)
        const bool expand_rows_on = gather_short && !(getenv("OSP_EXPAND_ROWS") && atoi(getenv("OSP_EXPAND_ROWS")) == 0);
        if (!expand_rows_on) sym_scatter_offsets_kernel<<<grid_for(nnz, 256), 256, 0, s>>>(perm, offs_sorted, rows_sorted, row_off, rw_cap, nnz, chunk_off);
        // (the product proper reads P together with the size of the result, merge_pipeline: one stream round trip less)
        if (partials_only || rowwise || row_sharded) P = d2h(offs_sorted + nnz, s);
        else P = kPartialsOnDevice;
        if (direct) {
            dsrc = DirectSrc{rowfirst, offs_sorted, bs_sorted, perm, b_colidx, chunk_off, direct_max};
            dsrc.b_rowptr = b_rowptr; dsrc.K = K; dsrc.nnz_b = (uint64_t)nnz_b; dsrc.keep = &sc;
            // gathered rows (osp_kernels.h): on unless OSP_GATHER=0 (debugging aid, A/B timing: every direct row is then written
            // by the multiply, as until round 4) or the row-wise variant runs (its tile kernel is another instantiation)
            dsrc.gather = gather_ok && gather_env >= 1;
            dsrc.expand_rows = expand_rows_on;
            if (expand_rows_on) { dsrc.chunk_off_ready = false; dsrc.rows_sorted = rows_sorted; dsrc.row_off = row_off; dsrc.rw_cap = rw_cap; dsrc.nnz = nnz; }
            dsrc.a_vals = with_av ? (const void *)av_sorted : (const void *)(a_vals + e0); dsrc.av_in_order = with_av; dsrc.b_vals = b_vals;
            if (dsrc.gather) {
                dsrc.gstat = (unsigned long long *)sc.get<uint64_t>(3);
                zero_async(s, {{dsrc.gstat, 3 * sizeof(uint64_t)}});
            }
        }
        if (gather_short) {
            // the short rows' chunks as run descriptors, chunks without entries left out (once per product)
            const ShortRunFlag sf{offs_sorted};
            uint32_t *cidx = ss.get<uint32_t>(nnz + 1), *cidx_tmp = ss.get<uint32_t>(scan_scratch_entries(nnz + 1));
            device_exclusive_scan<ShortRunFlag, uint32_t>(sf, nnz, cidx, cidx_tmp, s);
            RunDesc<T> *runs0 = sc.get<RunDesc<T>>(nnz);   // (bound: every chunk; the count stays on the device)
            uint32_t *rowfirst0 = sc.get<uint32_t>(M + 1);
            short_runs_kernel<T><<<grid_for(nnz, 256), 256, 0, s>>>(sf, cidx, nnz, bs_sorted, av_sorted, runs0);
            short_rowfirst_kernel<<<grid_for(M + 1, 256), 256, 0, s>>>(rowfirst, cidx, M, rowfirst0);
            srun = ShortRuns<T>{runs0, rowfirst0, b_colidx, b_vals};
        }
        if (rowwise) {
            ct = ChunkTable<T>{offs_sorted, bs_sorted, perm, rowfirst, a_vals + e0, b_colidx, b_vals, (uint32_t)rw_cap, 1u};
            uint32_t *flag_scan = ss.get<uint32_t>(M + 1);
            device_exclusive_scan<HeavyRowFlag, uint32_t>(HeavyRowFlag{row_off, 0, (uint32_t)rw_cap}, M, flag_scan, (uint32_t *)scan_tmp, s);
            n_long_rows = d2h(flag_scan + M, s);
        }
    }
    tm.end(PH_SYM);
    res->info.partials = P;

    OuterProducer<T> prod;
    prod.ctx = ctx; prod.res = res;
    prod.a_colptr = a_colptr; prod.a_rowidx = a_rowidx; prod.a_vals = a_vals;
    prod.b_rowptr = b_rowptr; prod.b_colidx = b_colidx; prod.b_vals = b_vals;
    prod.k0 = k0; prod.k1 = k1; prod.e0 = e0; prod.chunk_off = chunk_off;
    const uint64_t nk = k1 - k0;
    prod.a_start = sc.get<int64_t>(nk); prod.a_cnt = sc.get<uint32_t>(nk);
    prod.prod = sc.get<uint64_t>(nk); prod.prod_off = sc.get<uint64_t>(nk + 1);
    prod.scan_tmp = sc.get<uint64_t>(scan_scratch_entries(nk));
    prod.nothing_staged = rowwise && n_long_rows == 0;
    prod.short_gathered = gather_short;
    if ((dsrc.gather || gather_short) && nnz && !partials_only) {
        prod.nnz = nnz;
        prod.kscan = sc.get<uint32_t>(nnz + 1);
        prod.kscan_tmp = sc.get<uint32_t>(scan_scratch_entries(std::max<uint64_t>(nnz, k1 - k0) + 1));
        prod.elist = sc.get<uint32_t>(nnz);
        prod.cscan = sc.get<uint32_t>(k1 - k0 + 1);
        prod.klist = sc.get<uint32_t>(k1 - k0 + 1);
    }

    if (partials_only) {
        // osp_spgemm_partials: the multiply phase alone; offsets and records belong to the result
        if (row_sharded) throw Error(OSP_ERR_ARG, "partial products of a row shard are not supported");
        res->partials = true;
        res->rowptr = (int64_t *)ctx->alloc((M + 1) * sizeof(int64_t));
        OSP_HIP(hipMemcpyAsync(res->rowptr, row_off, (M + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
        res->vals = ctx->alloc(std::max<uint64_t>(P, 1) * sizeof(Part<T>));
        res->info.panels = 1;
        res->info.nnz_c = P;
        res->info.row_begin = 0;
        res->info.row_end = M;
        tm.begin(PH_MUL);
        if (P) prod.produce(0, M, true, 0, P, (Part<T> *)res->vals, tm, nullptr, nullptr);
        tm.end(PH_MUL);
        OSP_HIP(hipEventRecord(ev.b, s));
        OSP_HIP(hipStreamSynchronize(s));
        res->info.ms_total = ev.ms();
        res->info.ms_symbolic = tm.total(PH_SYM);
        res->info.ms_multiply = tm.total(PH_MUL);
        res->info.ms_multiply_kernel = tm.total(PH_MUL_K);
        return;
    }
    // row-sharded: A holds this rank's rows only, so the staging offsets start at 0 at r_lo and P is the shard's count
    const uint64_t off_lo = 0, P_rows = P;
    merge_pipeline<T>(ctx, res, prod, M, N, row_off, P_rows, cfg.partial_capacity, tm, r_lo, r_hi, off_lo, sink,
                      rowwise ? &ct : nullptr, (direct && nnz) ? &dsrc : nullptr, nullptr, gather_short ? &srun : nullptr);

    OSP_HIP(hipEventRecord(ev.b, s));
    OSP_HIP(hipStreamSynchronize(s));
    const float ms = ev.ms();
    if (gather_short) res->info.gathered_short_partials = res->info.partials - res->info.heavy_partials;
    if (dsrc.gstat && res->info.direct_rows) {
        uint64_t gs[3] = {0, 0, 0};
        copy_d2h(gs, dsrc.gstat, sizeof(gs), s);
        res->info.gathered_rows = gs[0]; res->info.gathered_partials = gs[1]; res->info.gathered_runs = gs[2];
    }
    if (getenv("OSP_VERBOSE")) {
        fprintf(stderr, "[osp] product done in %.1f ms; pool misses so far: %llu hipMalloc calls, %.1f GB, %.1f ms\n", ms,
                (unsigned long long)ctx->malloc_calls, ctx->malloc_bytes / 1e9, ctx->malloc_ms);
    }
    res->info.ms_total = ms;
    res->info.ms_symbolic = tm.total(PH_SYM);
    res->info.ms_multiply = tm.total(PH_MUL);
    res->info.ms_merge = tm.total(PH_MERGE);
    res->info.ms_compact = tm.total(PH_COMPACT);
    res->info.ms_multiply_kernel = tm.total(PH_MUL_K);
    res->info.ms_merge_kernel = tm.total(PH_MERGE_K);
    res->info.ms_split_kernel = tm.total(PH_SPLIT_K);
    res->info.ms_direct_plan_kernel = tm.total(PH_PLAN_K);
    res->info.ms_hub_plan_kernel = tm.total(PH_HUB_K);
    res->info.ms_expand_kernel = tm.total(PH_EXPAND_K);
}

// COO (device arrays, any order) -> compressed by `seg` with ascending `inner` indices; all outputs in `sc`.
template <class T>
static void coo_to_compressed_device(Context *ctx, Scratch &sc, uint64_t nseg, uint64_t ninner, uint64_t nnz, const uint32_t *seg,
                                     const uint32_t *inner, const T *vals, const char *what, int64_t **ptr_out,
                                     uint32_t **idx_out, T **val_out) {
    hipStream_t s = ctx->stream;
    int64_t *ptr = sc.get<int64_t>(nseg + 1);
    uint32_t *idx = sc.get<uint32_t>(nnz);
    T *ov = sc.get<T>(nnz);
    *ptr_out = ptr; *idx_out = idx; *val_out = ov;
    if (nnz == 0) {
        OSP_HIP(hipMemsetAsync(ptr, 0, (nseg + 1) * sizeof(int64_t), s));
        return;
    }
    Scratch ss(ctx);
    uint32_t *ka = ss.get<uint32_t>(nnz), *pa = ss.get<uint32_t>(nnz), *kb = ss.get<uint32_t>(nnz), *pb = ss.get<uint32_t>(nnz);
    uint32_t *k1 = ss.get<uint32_t>(nnz), *perm1 = ss.get<uint32_t>(nnz), *k2 = ss.get<uint32_t>(nnz);
    uint32_t *seg_sorted = ss.get<uint32_t>(nnz), *perm2 = ss.get<uint32_t>(nnz);
    uint32_t *hist = ss.get<uint32_t>(rs_hist_entries(nnz));
    uint32_t *hist_tmp = ss.get<uint32_t>(scan_scratch_entries(rs_hist_entries(nnz)));
    uint32_t *flags = ss.get<uint32_t>(1);
    OSP_HIP(hipMemsetAsync(flags, 0, sizeof(uint32_t), s));
    // stable LSD: by inner index first, then by segment
    device_sort_rows<RsStoreEpilogue>(inner, nnz, std::max(1, bits_for(ninner)), ka, pa, kb, pb, hist, hist_tmp,
                                      RsStoreEpilogue{k1, perm1}, s, ctx->rank_atomic);
    ingest_gather_u32_kernel<<<grid_for(nnz, 256), 256, 0, s>>>(seg, perm1, nnz, k2);
    device_sort_rows<RsStoreEpilogue>(k2, nnz, std::max(1, bits_for(nseg)), ka, pa, kb, pb, hist, hist_tmp,
                                      RsStoreEpilogue{seg_sorted, perm2}, s, ctx->rank_atomic, perm1);
    ingest_finish_kernel<T><<<grid_for(nnz, 256), 256, 0, s>>>(seg_sorted, perm2, inner, vals, nnz, nseg, ninner, idx, ov, flags);
    ingest_ptr_kernel<<<grid_for(nseg + 1, 256), 256, 0, s>>>(seg_sorted, nnz, nseg, ptr);
    check_flags(d2h(flags, s), what);
}

template <class T>
static void spgemm_coo_impl(Context *ctx, Result *res, uint64_t M, uint64_t K, uint64_t N, uint64_t nnz_a, const uint32_t *a_rows,
                            const uint32_t *a_cols, const T *a_vals, uint64_t nnz_b, const uint32_t *b_rows, const uint32_t *b_cols,
                            const T *b_vals, osp_memspace_t space, const osp_config_t &cfg) {
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    EventPair ev;
    OSP_HIP(hipEventRecord(ev.a, s));
    const uint32_t *ar = to_device(sc, a_rows, nnz_a, space, s), *ac = to_device(sc, a_cols, nnz_a, space, s);
    const uint32_t *br = to_device(sc, b_rows, nnz_b, space, s), *bc = to_device(sc, b_cols, nnz_b, space, s);
    const T *av = to_device(sc, a_vals, nnz_a, space, s), *bv = to_device(sc, b_vals, nnz_b, space, s);
    int64_t *ap, *bp;
    uint32_t *ai, *bi;
    T *acv, *bcv;
    coo_to_compressed_device<T>(ctx, sc, K, M, nnz_a, ac, ar, av, "A (COO)", &ap, &ai, &acv);  // csc = coo2csr<true>(A, K)
    coo_to_compressed_device<T>(ctx, sc, K, N, nnz_b, br, bc, bv, "B (COO)", &bp, &bi, &bcv);  // csr = coo2csr(B, K)
    OSP_HIP(hipEventRecord(ev.b, s));
    osp_config_t c2 = cfg;
    c2.validate = 0;  // ordering, ranges and duplicates were just established
    spgemm_impl<T>(ctx, res, M, K, N, ap, ai, acv, bp, bi, bcv, OSP_DEVICE, c2);
    const float ms = ev.ms();
    res->info.ms_ingest = ms;
    res->info.ms_total += ms;
}

// The reference's in-memory operands as they stand (osp_spgemm_csc_csr_aos): packed {u32 idx; T val} records -- the
// layout of Part<T> -- are split into index and value arrays on the device.
template <class T>
__global__ void aos_unpack_kernel(const Part<T> *__restrict__ data, uint64_t n, uint32_t *__restrict__ idx, T *__restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PartWords<T> r = load_part_words(data + i);
    idx[i] = r.col();
    val[i] = r.val();
}
template <class T>
static void spgemm_aos_impl(Context *ctx, Result *res, uint64_t M, uint64_t K, uint64_t N, const uint64_t *a_pos, const void *a_data,
                            const uint64_t *b_pos, const void *b_data, osp_memspace_t space, const osp_config_t &cfg) {
    static_assert(sizeof(Part<T>) == 4 + sizeof(T), "Part<T> must be the reference's packed CSRElement (common.h:10-16)");
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    // size_t offsets are taken as int64 (same bits below 2^63; larger values fail the pointer check)
    const int64_t *ap = to_device(sc, (const int64_t *)a_pos, K + 1, space, s);
    const int64_t *bp = to_device(sc, (const int64_t *)b_pos, K + 1, space, s);
    int64_t nnz_a, nnz_b;
    if (space == OSP_HOST) { nnz_a = (int64_t)a_pos[K]; nnz_b = (int64_t)b_pos[K]; }
    else { nnz_a = d2h(ap + K, s); nnz_b = d2h(bp + K, s); }
    if (nnz_a < 0 || nnz_b < 0 || (uint64_t)nnz_a >= 0xffffffffull || (uint64_t)nnz_b >= 0xffffffffull)
        throw Error(OSP_ERR_ARG, "operands with >= 2^32 non-zeros are not supported");
    if ((nnz_a && !a_data) || (nnz_b && !b_data)) throw Error(OSP_ERR_ARG, "null data array");
    const Part<T> *ad = to_device(sc, (const Part<T> *)a_data, (uint64_t)nnz_a, space, s);
    const Part<T> *bd = to_device(sc, (const Part<T> *)b_data, (uint64_t)nnz_b, space, s);
    uint32_t *ai = sc.get<uint32_t>(nnz_a), *bi = sc.get<uint32_t>(nnz_b);
    T *av = sc.get<T>(nnz_a), *bv = sc.get<T>(nnz_b);
    if (nnz_a) aos_unpack_kernel<T><<<grid_for(nnz_a, 256), 256, 0, s>>>(ad, (uint64_t)nnz_a, ai, av);
    if (nnz_b) aos_unpack_kernel<T><<<grid_for(nnz_b, 256), 256, 0, s>>>(bd, (uint64_t)nnz_b, bi, bv);
    spgemm_impl<T>(ctx, res, M, K, N, ap, ai, av, bp, bi, bv, OSP_DEVICE, cfg);
}

template <class T>
static void merge_parts_impl(Context *ctx, Result *res, uint64_t M, uint64_t N, int nparts,
                             const int64_t *const *rowptrs, const uint32_t *const *colidxs, const void *const *valss,
                             osp_memspace_t space, const osp_config_t &cfg) {
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    PhaseTimer tm(s);
    EventPair ev;
    OSP_HIP(hipEventRecord(ev.a, s));
    std::vector<const int64_t *> rp(nparts);
    std::vector<const uint32_t *> ci(nparts);
    std::vector<const T *> va(nparts);
    uint64_t nnz_in = 0;
    for (int p = 0; p < nparts; p++) {
        rp[p] = to_device(sc, rowptrs[p], M + 1, space, s);
        int64_t nnz = (space == OSP_HOST) ? rowptrs[p][M] : d2h(rp[p] + M, s);
        if (nnz < 0) throw Error(OSP_ERR_ARG, "negative nnz in part");
        ci[p] = to_device(sc, colidxs[p], nnz, space, s);
        va[p] = to_device(sc, (const T *)valss[p], nnz, space, s);
        nnz_in += nnz;
    }
    res->info.nnz_a = nnz_in;
    if (getenv("OSP_VERBOSE"))
        fprintf(stderr, "[osp] merge_csr_parts M=%llu N=%llu parts=%d entries=%llu %s operands\n", (unsigned long long)M,
                (unsigned long long)N, nparts, (unsigned long long)nnz_in, space == OSP_HOST ? "host" : "device");
    const int64_t **d_rp = (const int64_t **)sc.get<void *>(nparts);
    const uint32_t **d_ci = (const uint32_t **)sc.get<void *>(nparts);
    const T **d_va = (const T **)sc.get<void *>(nparts);
    copy_h2d(d_rp, rp.data(), nparts * sizeof(void *), s);
    copy_h2d(d_ci, ci.data(), nparts * sizeof(void *), s);
    copy_h2d(d_va, va.data(), nparts * sizeof(void *), s);
    tm.begin(PH_SYM);
    const uint64_t ncand = M * (uint64_t)nparts;  // candidate chunk (r, p) = row r of part p
    uint64_t *row_off = sc.get<uint64_t>(M + 1);
    uint64_t *offs = sc.get<uint64_t>(ncand + 1);
    uint64_t *scan_tmp = sc.get<uint64_t>(scan_scratch_entries(ncand));
    const PartsChunkLen plen{d_rp, nparts};
    device_exclusive_scan<PartsChunkLen, uint64_t>(plen, ncand, offs, scan_tmp, s);
    parts_rows_kernel<<<grid_for(M + 1, 256), 256, 0, s>>>(offs, nparts, M, row_off);
    const uint64_t P = d2h(offs + ncand, s);
    tm.end(PH_SYM);
    res->info.partials = P;
    PartsProducer<T> prod;
    prod.ctx = ctx; prod.d_rowptrs = d_rp; prod.d_colidxs = d_ci; prod.d_valss = d_va;
    prod.nparts = nparts; prod.row_off = row_off;
    merge_pipeline<T>(ctx, res, prod, M, N, row_off, P, cfg.partial_capacity, tm);
    OSP_HIP(hipEventRecord(ev.b, s));
    OSP_HIP(hipStreamSynchronize(s));
    res->info.ms_total = ev.ms();
    res->info.ms_symbolic = tm.total(PH_SYM);
    res->info.ms_multiply = tm.total(PH_MUL);
    res->info.ms_merge = tm.total(PH_MERGE);
    res->info.ms_compact = tm.total(PH_COMPACT);
    res->info.ms_multiply_kernel = tm.total(PH_MUL_K);
    res->info.ms_merge_kernel = tm.total(PH_MERGE_K);
    res->info.ms_split_kernel = tm.total(PH_SPLIT_K);
}

template <class T>
static void merge_record_parts_impl(Context *ctx, Result *res, uint64_t M, uint64_t N, int nparts, const int64_t *const *rowptrs,
                                    const void *const *records, osp_memspace_t space, const osp_config_t &cfg,
                                    const std::vector<uint64_t> *cuts = nullptr,
                                    const std::function<void(uint64_t, uint64_t)> *before = nullptr) {
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    PhaseTimer tm(s);
    EventPair ev;
    OSP_HIP(hipEventRecord(ev.a, s));
    std::vector<const int64_t *> rp(nparts);
    std::vector<const Part<T> *> rc(nparts);
    uint64_t nnz_in = 0;
    for (int p = 0; p < nparts; p++) {
        rp[p] = to_device(sc, rowptrs[p], M + 1, space, s);
        const int64_t n = (space == OSP_HOST) ? rowptrs[p][M] : d2h(rp[p] + M, s);
        if (n < 0) throw Error(OSP_ERR_ARG, "negative record count in part");
        rc[p] = to_device(sc, (const Part<T> *)records[p], (uint64_t)n, space, s);
        nnz_in += (uint64_t)n;
    }
    res->info.nnz_a = nnz_in;
    if (cfg.validate) {
        // offsets monotone from 0 to the record count, columns below N: what the split and dense paths index with
        uint32_t *flags = sc.get<uint32_t>(1);
        OSP_HIP(hipMemsetAsync(flags, 0, sizeof(uint32_t), s));
        for (int p = 0; p < nparts; p++) {
            const uint64_t n = (uint64_t)((space == OSP_HOST) ? rowptrs[p][M] : d2h(rp[p] + M, s));
            validate_ptr_kernel<<<grid_for(M + 1, 256), 256, 0, s>>>(rp[p], M, n, flags);
            if (n) validate_record_cols_kernel<T><<<grid_for(n, 256), 256, 0, s>>>(rc[p], n, N, flags);
        }
        check_flags(d2h(flags, s), "record parts");
    }
    const int64_t **d_rp = (const int64_t **)sc.get<void *>(nparts);
    const Part<T> **d_rc = (const Part<T> **)sc.get<void *>(nparts);
    copy_h2d(d_rp, rp.data(), nparts * sizeof(void *), s);
    copy_h2d(d_rc, rc.data(), nparts * sizeof(void *), s);
    tm.begin(PH_SYM);
    const uint64_t ncand = M * (uint64_t)nparts;  // candidate chunk (r, p) = row r of part p
    uint64_t *row_off = sc.get<uint64_t>(M + 1);
    uint64_t *offs = sc.get<uint64_t>(ncand + 1);
    uint64_t *scan_tmp = sc.get<uint64_t>(scan_scratch_entries(ncand));
    device_exclusive_scan<PartsChunkLen, uint64_t>(PartsChunkLen{d_rp, nparts}, ncand, offs, scan_tmp, s);
    parts_rows_kernel<<<grid_for(M + 1, 256), 256, 0, s>>>(offs, nparts, M, row_off);
    const uint64_t P = d2h(offs + ncand, s);
    tm.end(PH_SYM);
    res->info.partials = P;
    RecordPartsProducer<T> prod;
    prod.ctx = ctx; prod.d_rowptrs = d_rp; prod.d_recs = d_rc; prod.nparts = nparts; prod.row_off = row_off; prod.before = before;
    merge_pipeline<T>(ctx, res, prod, M, N, row_off, P, cfg.partial_capacity, tm, 0, ~0ull, 0, nullptr, nullptr, nullptr, cuts);
    OSP_HIP(hipEventRecord(ev.b, s));
    OSP_HIP(hipStreamSynchronize(s));
    res->info.ms_total = ev.ms();
    res->info.ms_symbolic = tm.total(PH_SYM);
    res->info.ms_multiply = tm.total(PH_MUL);
    res->info.ms_merge = tm.total(PH_MERGE);
    res->info.ms_compact = tm.total(PH_COMPACT);
    res->info.ms_merge_kernel = tm.total(PH_MERGE_K);
    res->info.ms_split_kernel = tm.total(PH_SPLIT_K);
}

// which of the two exact variants of the order-sensitive steps this context runs (visible in every result)
static void note_variants(const Context *ctx, Result *res) {
    res->info.rank_atomic = ctx->rank_atomic ? 1u : 0u;
    res->info.dense_atomic = ctx->dense_atomic[res->dtype == OSP_F64] ? 1u : 0u;
}

// relu(C + bias) with the zeros dropped, as a new CSR (osp_epilogue.h)
template <class T>
static void bias_relu_impl(Context *ctx, const Result *in, Result *res, const T *bias_in, osp_memspace_t bias_space, int relu) {
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    EventPair ev;
    OSP_HIP(hipEventRecord(ev.a, s));
    const uint64_t M = in->info.M, N = in->info.N;
    const T *bias = bias_in ? to_device(sc, bias_in, N, bias_space, s) : nullptr;
    res->info = in->info;
    res->rowptr = (int64_t *)ctx->alloc((M + 1) * sizeof(int64_t));
    uint32_t *cnt = sc.get<uint32_t>(M + 1);
    uint64_t *tmp = sc.get<uint64_t>(scan_scratch_entries(M + 1));
    const unsigned grid = grid_for(std::max<uint64_t>(M, 1) * kWave, 256);
    bias_relu_rows_kernel<T, false><<<grid, 256, 0, s>>>(in->rowptr, in->colidx, (const T *)in->vals, M, N, bias, relu, cnt, nullptr, nullptr, nullptr);
    device_exclusive_scan<LoadU32As64, uint64_t>(LoadU32As64{cnt}, M, (uint64_t *)res->rowptr, tmp, s);
    const uint64_t nnz = (uint64_t)d2h(res->rowptr + M, s);
    res->colidx = (uint32_t *)ctx->alloc(std::max<uint64_t>(nnz, 1) * sizeof(uint32_t));
    res->vals = ctx->alloc(std::max<uint64_t>(nnz, 1) * sizeof(T));
    if (nnz)
        bias_relu_rows_kernel<T, true><<<grid, 256, 0, s>>>(in->rowptr, in->colidx, (const T *)in->vals, M, N, bias, relu, nullptr, res->rowptr, res->colidx,
                                                            (T *)res->vals);
    OSP_HIP(hipEventRecord(ev.b, s));
    OSP_HIP(hipStreamSynchronize(s));
    OSP_HIP(hipGetLastError());
    res->info.nnz_c = nnz;
    res->info.ms_total = ev.ms();
}

static void destroy_result(Result *r) {
    if (!r) return;
    if (r->ctx) {
        r->ctx->release(r->rowptr);
        r->ctx->release(r->colidx);
        r->ctx->release(r->vals);
    }
    delete r;
}

}  // namespace osp

#include "osp_multi.h"

using namespace osp;

// ---- C ABI ---------------------------------------------------------------------------------------
#define OSP_GUARD_BEGIN try {
#define OSP_GUARD_END                                                 \
    }                                                                 \
    catch (const Error &e) { return fail(e.status, "%s", e.what()); } \
    catch (const std::bad_alloc &) { return fail(OSP_ERR_ALLOC, "host allocation failed"); } \
    catch (const std::exception &e) { return fail(OSP_ERR_HIP, "%s", e.what()); }

extern "C" {

const char *osp_status_string(int st) {
    switch (st) {
        case OSP_OK: return "ok";
        case OSP_ERR_DIM: return "inner dimensions differ";
        case OSP_ERR_ARG: return "bad argument";
        case OSP_ERR_ALLOC: return "allocation failed";
        case OSP_ERR_HIP: return "HIP error";
        case OSP_ERR_IO: return "I/O error";
        case OSP_ERR_RANGE: return "index out of range";
        case OSP_ERR_CAPACITY: return "staging capacity exceeded";
        case OSP_ERR_UNSORTED: return "indices not ascending";
        case OSP_ERR_DUPLICATE: return "duplicate coordinate (233)";
        default: return "unknown status";
    }
}

void osp_config_default(osp_config_t *cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->validate = 1;
}

static int context_create(int device, void *stream, bool own, osp_context_t *out) {
    if (!out) return fail(OSP_ERR_ARG, "null context pointer");
    OSP_GUARD_BEGIN
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw Error(OSP_ERR_HIP, "no HIP device visible: this library has no CPU path");
    if (device < 0 || device >= ndev) throw Error(OSP_ERR_ARG, "device ordinal out of range");
    OSP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    OSP_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        throw Error(OSP_ERR_HIP, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    Context *c = new Context;
    c->device = device;
    c->cus = (uint32_t)prop.multiProcessorCount;
    if (own) { OSP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    else c->stream = (hipStream_t)stream;
    try {
        // Which stable rank this context uses.  The atomic one needs a property of LDS atomics that is not documented, so
        // it is tested here on THIS device; when the test fails the context falls back to the ballot instantiations of the
        // same kernels (slower -- merge +20 % -- and just as exact) instead of refusing to work.
        const char *force = getenv("OSP_RANK");   // "ballot" | "atomic": debugging and tests/test_gpu_parity.py
        if (force && strcmp(force, "ballot") == 0) c->rank_atomic = false;
        else if (force && strcmp(force, "atomic") == 0) c->rank_atomic = true;
        if (c->rank_atomic) {
            Scratch sc(c);
            uint32_t *bad = sc.get<uint32_t>(1);
            OSP_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), c->stream));
            rank_order_selftest_kernel<<<64, 256, 0, c->stream>>>(bad);
            if (d2h(bad, c->stream) != 0) {
                c->rank_atomic = false;
                if (getenv("OSP_VERBOSE")) fprintf(stderr, "[osp] LDS atomics do not return old values in lane order on this device: using ballot ranks\n");
            }
        }
        const char *fadd = getenv("OSP_DENSE_ADD");
        for (int wide = 0; wide < 2; wide++) {
            if (fadd && strcmp(fadd, "ballot") == 0) c->dense_atomic[wide] = false;
            else if (fadd && strcmp(fadd, "atomic") == 0) c->dense_atomic[wide] = true;
            if (!c->dense_atomic[wide]) continue;
            Scratch sc(c);
            uint32_t *bad = sc.get<uint32_t>(1);
            OSP_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), c->stream));
            if (wide) fadd_order_selftest_kernel<double><<<64, 256, 0, c->stream>>>(bad);
            else fadd_order_selftest_kernel<float><<<64, 256, 0, c->stream>>>(bad);
            if (d2h(bad, c->stream) != 0) {
                c->dense_atomic[wide] = false;
                if (getenv("OSP_VERBOSE"))
                    fprintf(stderr, "[osp] LDS %s atomics do not add in lane order (or flush subnormals) on this device: dense segments by ballot ranks\n",
                            wide ? "f64" : "f32");
            }
        }
    } catch (...) {
        c->trim();
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        throw;
    }
    *out = (osp_context_t)c;
    return OSP_OK;
    OSP_GUARD_END
}
int osp_context_create(int device, osp_context_t *ctx) { return context_create(device, nullptr, true, ctx); }
int osp_context_create_on_stream(int device, void *hip_stream, osp_context_t *ctx) {
    return context_create(device, hip_stream, false, ctx);
}
int osp_context_trim(osp_context_t c) {
    if (!c) return fail(OSP_ERR_ARG, "null context");
    ((Context *)c)->trim();
    return OSP_OK;
}
int osp_context_alloc(osp_context_t c_, uint64_t bytes, void **device_ptr) {
    Context *c = (Context *)c_;
    if (!c || !device_ptr) return fail(OSP_ERR_ARG, "null argument");
    OSP_GUARD_BEGIN
    OSP_HIP(hipSetDevice(c->device));
    *device_ptr = c->alloc((size_t)bytes);
    return OSP_OK;
    OSP_GUARD_END
}
int osp_context_free(osp_context_t c_, void *device_ptr) {
    Context *c = (Context *)c_;
    if (!c) return fail(OSP_ERR_ARG, "null context");
    c->release(device_ptr);
    return OSP_OK;
}
int osp_context_destroy(osp_context_t c_) {
    Context *c = (Context *)c_;
    if (!c) return OSP_OK;
    if (c->sibling) c->sibling->sibling = nullptr;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->trim();
    for (auto &kv : c->live) (void)hipFree(kv.first);
    c->drop_aux();
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return OSP_OK;
}

int osp_spgemm_csc_csr(osp_context_t ctx_, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                       const int64_t *a_colptr, const uint32_t *a_rowidx, const void *a_vals,
                       const int64_t *b_rowptr, const uint32_t *b_colidx, const void *b_vals,
                       osp_memspace_t space, const osp_config_t *cfg_, osp_result_t *result) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !result) return fail(OSP_ERR_ARG, "null context or result pointer");
    if (!a_colptr || !b_rowptr) return fail(OSP_ERR_ARG, "null pointer array");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (space != OSP_HOST && space != OSP_DEVICE) return fail(OSP_ERR_ARG, "bad memory space");
    if (M >= 0xffffffffull || N > 0xffffffffull || K >= 0xffffffffull) return fail(OSP_ERR_ARG, "dimension exceeds the u32 index type");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    Result *res = new Result;
    res->ctx = ctx;
    res->dtype = dtype;
    res->info.M = M; res->info.K = K; res->info.N = N; res->info.dtype = dtype;
    note_variants(ctx, res);
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (dtype == OSP_F32)
            spgemm_impl<float>(ctx, res, M, K, N, a_colptr, a_rowidx, (const float *)a_vals, b_rowptr, b_colidx,
                               (const float *)b_vals, space, cfg);
        else
            spgemm_impl<double>(ctx, res, M, K, N, a_colptr, a_rowidx, (const double *)a_vals, b_rowptr, b_colidx,
                                (const double *)b_vals, space, cfg);
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *result = (osp_result_t)res;
    return OSP_OK;
}

int osp_spgemm_csc_csr_aos(osp_context_t ctx_, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N, const uint64_t *a_pos,
                           const void *a_data, const uint64_t *b_pos, const void *b_data, osp_memspace_t space,
                           const osp_config_t *cfg_, osp_result_t *result) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !result) return fail(OSP_ERR_ARG, "null context or result pointer");
    if (!a_pos || !b_pos) return fail(OSP_ERR_ARG, "null pointer array");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (space != OSP_HOST && space != OSP_DEVICE) return fail(OSP_ERR_ARG, "bad memory space");
    if (M >= 0xffffffffull || N > 0xffffffffull || K >= 0xffffffffull) return fail(OSP_ERR_ARG, "dimension exceeds the u32 index type");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    Result *res = new Result;
    res->ctx = ctx;
    res->dtype = dtype;
    res->info.M = M; res->info.K = K; res->info.N = N; res->info.dtype = dtype;
    note_variants(ctx, res);
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (dtype == OSP_F32) spgemm_aos_impl<float>(ctx, res, M, K, N, a_pos, a_data, b_pos, b_data, space, cfg);
        else spgemm_aos_impl<double>(ctx, res, M, K, N, a_pos, a_data, b_pos, b_data, space, cfg);
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *result = (osp_result_t)res;
    return OSP_OK;
}

int osp_spgemm_csc_csr_panels(osp_context_t ctx_, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                              const int64_t *a_colptr, const uint32_t *a_rowidx, const void *a_vals,
                              const int64_t *b_rowptr, const uint32_t *b_colidx, const void *b_vals,
                              osp_memspace_t space, const osp_config_t *cfg_, osp_panel_fn fn, void *user,
                              osp_result_info_t *info) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !fn) return fail(OSP_ERR_ARG, "null context or panel callback");
    if (!a_colptr || !b_rowptr) return fail(OSP_ERR_ARG, "null pointer array");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (space != OSP_HOST && space != OSP_DEVICE) return fail(OSP_ERR_ARG, "bad memory space");
    if (M >= 0xffffffffull || N > 0xffffffffull || K >= 0xffffffffull) return fail(OSP_ERR_ARG, "dimension exceeds the u32 index type");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    Result *res = new Result;  // carries the counters only: no output arrays are attached in streaming mode
    res->ctx = ctx;
    res->dtype = dtype;
    res->info.M = M; res->info.K = K; res->info.N = N; res->info.dtype = dtype;
    note_variants(ctx, res);
    const PanelSink sink{fn, user};
    int st = OSP_OK;
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (dtype == OSP_F32)
            spgemm_impl<float>(ctx, res, M, K, N, a_colptr, a_rowidx, (const float *)a_vals, b_rowptr, b_colidx,
                               (const float *)b_vals, space, cfg, &sink);
        else
            spgemm_impl<double>(ctx, res, M, K, N, a_colptr, a_rowidx, (const double *)a_vals, b_rowptr, b_colidx,
                                (const double *)b_vals, space, cfg, &sink);
        if (info) *info = res->info;
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        st = fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        st = fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    destroy_result(res);
    return st;
}

int osp_spgemm_coo(osp_context_t ctx_, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N, uint64_t nnz_a,
                   const uint32_t *a_rows, const uint32_t *a_cols, const void *a_vals, uint64_t nnz_b,
                   const uint32_t *b_rows, const uint32_t *b_cols, const void *b_vals, osp_memspace_t space,
                   const osp_config_t *cfg_, osp_result_t *result) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !result) return fail(OSP_ERR_ARG, "null context or result pointer");
    if ((nnz_a && (!a_rows || !a_cols || !a_vals)) || (nnz_b && (!b_rows || !b_cols || !b_vals))) return fail(OSP_ERR_ARG, "null operand array");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (space != OSP_HOST && space != OSP_DEVICE) return fail(OSP_ERR_ARG, "bad memory space");
    if (M >= 0xffffffffull || N > 0xffffffffull || K >= 0xffffffffull || nnz_a >= 0xffffffffull || nnz_b >= 0xffffffffull)
        return fail(OSP_ERR_ARG, "dimension or nnz exceeds the u32 index type");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    Result *res = new Result;
    res->ctx = ctx;
    res->dtype = dtype;
    res->info.M = M; res->info.K = K; res->info.N = N; res->info.dtype = dtype;
    note_variants(ctx, res);
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (dtype == OSP_F32)
            spgemm_coo_impl<float>(ctx, res, M, K, N, nnz_a, a_rows, a_cols, (const float *)a_vals, nnz_b, b_rows, b_cols,
                                   (const float *)b_vals, space, cfg);
        else
            spgemm_coo_impl<double>(ctx, res, M, K, N, nnz_a, a_rows, a_cols, (const double *)a_vals, nnz_b, b_rows, b_cols,
                                    (const double *)b_vals, space, cfg);
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *result = (osp_result_t)res;
    return OSP_OK;
}

int osp_merge_csr_parts(osp_context_t ctx_, osp_dtype_t dtype, uint64_t M, uint64_t N, int nparts,
                        const int64_t *const *rowptrs, const uint32_t *const *colidxs,
                        const void *const *valss, osp_memspace_t space, const osp_config_t *cfg_,
                        osp_result_t *result) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !result || !rowptrs || !colidxs || !valss) return fail(OSP_ERR_ARG, "null argument");
    if (nparts < 1) return fail(OSP_ERR_ARG, "nparts must be >= 1");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (M >= 0xffffffffull || N > 0xffffffffull) return fail(OSP_ERR_ARG, "dimension exceeds the u32 index type");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    Result *res = new Result;
    res->ctx = ctx;
    res->dtype = dtype;
    res->info.M = M; res->info.N = N; res->info.dtype = dtype;
    note_variants(ctx, res);
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (dtype == OSP_F32) merge_parts_impl<float>(ctx, res, M, N, nparts, rowptrs, colidxs, valss, space, cfg);
        else merge_parts_impl<double>(ctx, res, M, N, nparts, rowptrs, colidxs, valss, space, cfg);
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *result = (osp_result_t)res;
    return OSP_OK;
}

int osp_spgemm_partials(osp_context_t ctx_, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N, const int64_t *a_colptr,
                        const uint32_t *a_rowidx, const void *a_vals, const int64_t *b_rowptr, const uint32_t *b_colidx,
                        const void *b_vals, osp_memspace_t space, const osp_config_t *cfg_, osp_result_t *result) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !result) return fail(OSP_ERR_ARG, "null context or result pointer");
    if (!a_colptr || !b_rowptr) return fail(OSP_ERR_ARG, "null pointer array");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (space != OSP_HOST && space != OSP_DEVICE) return fail(OSP_ERR_ARG, "bad memory space");
    if (M >= 0xffffffffull || N > 0xffffffffull || K >= 0xffffffffull) return fail(OSP_ERR_ARG, "dimension exceeds the u32 index type");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    Result *res = new Result;
    res->ctx = ctx;
    res->dtype = dtype;
    res->info.M = M; res->info.K = K; res->info.N = N; res->info.dtype = dtype;
    note_variants(ctx, res);
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (dtype == OSP_F32)
            spgemm_impl<float>(ctx, res, M, K, N, a_colptr, a_rowidx, (const float *)a_vals, b_rowptr, b_colidx, (const float *)b_vals,
                               space, cfg, nullptr, true);
        else
            spgemm_impl<double>(ctx, res, M, K, N, a_colptr, a_rowidx, (const double *)a_vals, b_rowptr, b_colidx,
                                (const double *)b_vals, space, cfg, nullptr, true);
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *result = (osp_result_t)res;
    return OSP_OK;
}

int osp_result_partials(osp_result_t r_, const int64_t **rowptr, const void **records) {
    Result *r = (Result *)r_;
    if (!r) return fail(OSP_ERR_ARG, "null result");
    if (!r->partials) return fail(OSP_ERR_ARG, "not a result of osp_spgemm_partials");
    if (rowptr) *rowptr = r->rowptr;
    if (records) *records = r->vals;
    return OSP_OK;
}

int osp_merge_record_parts(osp_context_t ctx_, osp_dtype_t dtype, uint64_t M, uint64_t N, int nparts, const int64_t *const *rowptrs,
                           const void *const *records, osp_memspace_t space, const osp_config_t *cfg_, osp_result_t *result) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !result || !rowptrs || !records) return fail(OSP_ERR_ARG, "null argument");
    if (nparts < 1) return fail(OSP_ERR_ARG, "nparts must be >= 1");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (space != OSP_HOST && space != OSP_DEVICE) return fail(OSP_ERR_ARG, "bad memory space");
    if (M >= 0xffffffffull || N > 0xffffffffull) return fail(OSP_ERR_ARG, "dimension exceeds the u32 index type");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    Result *res = new Result;
    res->ctx = ctx;
    res->dtype = dtype;
    res->info.M = M; res->info.N = N; res->info.dtype = dtype;
    note_variants(ctx, res);
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (dtype == OSP_F32) merge_record_parts_impl<float>(ctx, res, M, N, nparts, rowptrs, records, space, cfg);
        else merge_record_parts_impl<double>(ctx, res, M, N, nparts, rowptrs, records, space, cfg);
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *result = (osp_result_t)res;
    return OSP_OK;
}

int osp_csr_bias_relu(osp_result_t in_, const void *bias, osp_memspace_t bias_space, int relu, osp_result_t *out) {
    Result *in = (Result *)in_;
    if (!in || !out) return fail(OSP_ERR_ARG, "null argument");
    if (in->partials) return fail(OSP_ERR_ARG, "a result of osp_spgemm_partials holds records, not a CSR");
    if (bias_space != OSP_HOST && bias_space != OSP_DEVICE) return fail(OSP_ERR_ARG, "bad memory space");
    Context *ctx = in->ctx;
    Result *res = new Result;
    res->ctx = ctx;
    res->dtype = in->dtype;
    try {
        OSP_HIP(hipSetDevice(ctx->device));
        if (in->dtype == OSP_F32) bias_relu_impl<float>(ctx, in, res, (const float *)bias, bias_space, relu);
        else bias_relu_impl<double>(ctx, in, res, (const double *)bias, bias_space, relu);
    } catch (const Error &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        (void)hipStreamSynchronize(ctx->stream);
        destroy_result(res);
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *out = (osp_result_t)res;
    return OSP_OK;
}

int osp_result_coo_rows(osp_result_t r_, uint32_t *rows_device) {
    Result *r = (Result *)r_;
    if (!r || !rows_device) return fail(OSP_ERR_ARG, "null argument");
    if (r->partials) return fail(OSP_ERR_ARG, "a result of osp_spgemm_partials holds records, not a CSR");
    OSP_GUARD_BEGIN
    OSP_HIP(hipSetDevice(r->ctx->device));
    const uint64_t M = r->info.M;
    if (M && r->info.nnz_c)
        csr_expand_rows_kernel<<<grid_for(M * kWave, 256), 256, 0, r->ctx->stream>>>(r->rowptr, M, rows_device);
    OSP_HIP(hipStreamSynchronize(r->ctx->stream));
    OSP_HIP(hipGetLastError());
    return OSP_OK;
    OSP_GUARD_END
}

// ---- what a plain stream reaches on this device (bench.py: roofline.peak_measured) ----
// 16 bytes per lane, one workgroup per 4 KB: the copy SURVEY.md 8d / BASELINE.md ask to be
// measured on the box beside the 8 TB/s of the data sheet (the reference prints its simulated DRAM rate,
// SimOuterSPACE.cpp:684-686).  rate = (bytes read + bytes written) / time.
int osp_stream_copy_probe(osp_context_t ctx_, uint64_t bytes, int reps, double *gbps) {
    Context *ctx = (Context *)ctx_;
    if (!ctx || !gbps) return fail(OSP_ERR_ARG, "null argument");
    if (bytes < 4096 || reps < 1 || reps > 1000) return fail(OSP_ERR_ARG, "bytes >= 4096, 1 <= reps <= 1000");
    OSP_GUARD_BEGIN
    OSP_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    const uint64_t n = bytes / sizeof(u32x4);
    u32x4 *src = sc.get<u32x4>(n), *dst = sc.get<u32x4>(n);
    OSP_HIP(hipMemsetAsync(src, 0x5a, n * sizeof(u32x4), s));
    if (n > 0xffffffffull * 256ull) return fail(OSP_ERR_ARG, "probe buffer too large");
    const unsigned grid = (unsigned)((n + 255) / 256);
    double best = 0;
    for (int nt = 0; nt < 2; nt++) {   // plain and non-temporal accesses: the better of the two is the measured roof
        for (int i = 0; i < 2; i++) {   // untimed: page tables, clocks
            if (nt) stream_copy_kernel<true><<<grid, 256, 0, s>>>(src, dst, n); else stream_copy_kernel<false><<<grid, 256, 0, s>>>(src, dst, n);
        }
        EventPair ev;
        OSP_HIP(hipEventRecord(ev.a, s));
        for (int i = 0; i < reps; i++) {
            if (nt) stream_copy_kernel<true><<<grid, 256, 0, s>>>(src, dst, n); else stream_copy_kernel<false><<<grid, 256, 0, s>>>(src, dst, n);
        }
        OSP_HIP(hipEventRecord(ev.b, s));
        OSP_HIP(hipStreamSynchronize(s));
        OSP_HIP(hipGetLastError());
        const double ms = ev.ms();
        if (ms > 0) best = std::max(best, 2.0 * (double)(n * sizeof(u32x4)) * reps / (ms * 1e-3) / 1e9);
    }
    *gbps = best;
    return OSP_OK;
    OSP_GUARD_END
}

// ---- several GPUs of one node (osp_multi.h) ----
int osp_multi_context_create(const int *devices, int ndev, osp_multi_context_t *out) {
    if (!devices || !out) return fail(OSP_ERR_ARG, "null argument");
    if (ndev < 1 || ndev > OSP_MULTI_MAX_RANKS) return fail(OSP_ERR_ARG, "between 1 and %d ranks", OSP_MULTI_MAX_RANKS);
    MultiContext *mc = new MultiContext;
    try {
        for (int g = 0; g < ndev; g++) {
            osp_context_t c = nullptr;
            const int st = osp_context_create(devices[g], &c);
            if (st) { delete mc; return st; }   // (the message is already set)
            mc->devices.push_back(devices[g]);
            mc->ctx.push_back((Context *)c);
            mc->copy.emplace_back((size_t)ndev, nullptr);
            mc->mctx.push_back(nullptr);
            osp_context_t c2 = nullptr;
            const int st2 = osp_context_create(devices[g], &c2);   // the merge of what arrives: a stream and a pool of its own
            if (st2) { delete mc; return st2; }
            mc->mctx.back() = (Context *)c2;
            mc->ctx.back()->sibling = mc->mctx.back();
            mc->mctx.back()->sibling = mc->ctx.back();
            OSP_HIP(hipSetDevice(devices[g]));
            for (int h = 0; h < ndev; h++) {
                if (h == g) continue;
                OSP_HIP(hipStreamCreateWithFlags(&mc->copy[g][h], hipStreamNonBlocking));   // one stream per destination: one per link
            }
        }
        // direct copies between the GPUs where the hardware allows them (xGMI); without peer access a copy is staged
        // through the host by the runtime, which is slower but correct
        for (int g = 0; g < ndev; g++)
            for (int h = 0; h < ndev; h++) {
                if (devices[g] == devices[h]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[g], devices[h]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[g]);
                    (void)hipDeviceEnablePeerAccess(devices[h], 0);
                    (void)hipGetLastError();   // "already enabled" is fine
                }
            }
    } catch (const Error &e) {
        delete mc;
        return fail(e.status, "%s", e.what());
    }
    *out = (osp_multi_context_t)mc;
    return OSP_OK;
}
int osp_multi_context_destroy(osp_multi_context_t mc) {
    delete (MultiContext *)mc;
    return OSP_OK;
}

int osp_multi_operands_create(osp_multi_context_t mc_, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N, const int64_t *a_colptr,
                              const uint32_t *a_rowidx, const void *a_vals, const int64_t *b_rowptr, const uint32_t *b_colidx,
                              const void *b_vals, osp_multi_operands_t *out) {
    MultiContext *mc = (MultiContext *)mc_;
    if (!mc || !out || !a_colptr || !b_rowptr) return fail(OSP_ERR_ARG, "null argument");
    if (dtype != OSP_F32 && dtype != OSP_F64) return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
    if (M >= 0xffffffffull || N > 0xffffffffull || K >= 0xffffffffull) return fail(OSP_ERR_ARG, "dimension exceeds the u32 index type");
    const int64_t nnz_a = a_colptr[K], nnz_b = b_rowptr[K];
    if (nnz_a < 0 || nnz_b < 0 || (uint64_t)nnz_a >= 0xffffffffull || (uint64_t)nnz_b >= 0xffffffffull)
        return fail(OSP_ERR_ARG, "operands with >= 2^32 non-zeros are not supported");
    if ((nnz_a && (!a_rowidx || !a_vals)) || (nnz_b && (!b_colidx || !b_vals))) return fail(OSP_ERR_ARG, "null operand array");
    for (uint64_t k = 0; k < K; k++)
        if (a_colptr[k + 1] < a_colptr[k] || b_rowptr[k + 1] < b_rowptr[k] || a_colptr[0] != 0 || b_rowptr[0] != 0)
            return fail(OSP_ERR_ARG, "pointer array is not a monotone 0..nnz sequence");
    MultiOperands *ops = new MultiOperands;
    ops->mc = mc; ops->dtype = dtype; ops->M = M; ops->K = K; ops->N = N;
    try {
        if (dtype == OSP_F32) multi_upload<float>(mc, ops, a_colptr, a_rowidx, (const float *)a_vals, b_rowptr, b_colidx, (const float *)b_vals);
        else multi_upload<double>(mc, ops, a_colptr, a_rowidx, (const double *)a_vals, b_rowptr, b_colidx, (const double *)b_vals);
    } catch (const Error &e) {
        delete ops;
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        delete ops;
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *out = (osp_multi_operands_t)ops;
    return OSP_OK;
}
int osp_multi_operands_destroy(osp_multi_operands_t ops) {
    delete (MultiOperands *)ops;
    return OSP_OK;
}

int osp_spgemm_multi(osp_multi_context_t mc_, osp_multi_operands_t ops_, const osp_config_t *cfg_, osp_multi_result_t *out) {
    MultiContext *mc = (MultiContext *)mc_;
    MultiOperands *ops = (MultiOperands *)ops_;
    if (!mc || !ops || !out) return fail(OSP_ERR_ARG, "null argument");
    if (ops->mc != mc) return fail(OSP_ERR_ARG, "operands belong to another multi-GPU context");
    osp_config_t cfg;
    if (cfg_) cfg = *cfg_; else osp_config_default(&cfg);
    MultiResult *res = new MultiResult;
    res->mc = mc;
    res->dtype = ops->dtype;
    try {
        if (cfg.validate) {
            // per slab, as the single-GPU entry point does: ordering, ranges, duplicates (233)
            for (size_t g = 0; g < mc->ctx.size(); g++) {
                Context *c = mc->ctx[g];
                OSP_HIP(hipSetDevice(c->device));
                const MultiOperands::Slab &sl = ops->slab[g];
                Scratch sc(c);
                uint32_t *flags = sc.get<uint32_t>(2);
                OSP_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(uint32_t), c->stream));
                if (sl.nnz_a) validate_idx_kernel<<<grid_for(sl.nnz_a, 256), 256, 0, c->stream>>>(sl.a_colptr, sl.a_rowidx, sl.K, sl.nnz_a, ops->M, flags);
                if (sl.nnz_b) validate_idx_kernel<<<grid_for(sl.nnz_b, 256), 256, 0, c->stream>>>(sl.b_rowptr, sl.b_colidx, sl.K, sl.nnz_b, ops->N, flags + 1);
                uint32_t fa = 0, fb = 0;
                { Gather gt(c->stream); gt.add(&fa, (const uint32_t *)flags); gt.add(&fb, (const uint32_t *)flags + 1); gt.wait(); }
                check_flags(fa, "A (CSC)");
                check_flags(fb, "B (CSR)");
            }
        }
        if (ops->dtype == OSP_F32) multi_product<float>(mc, ops, res, cfg);
        else multi_product<double>(mc, ops, res, cfg);
        res->info.ms_upload = ops->ms_upload;
    } catch (const Error &e) {
        delete res;
        return fail(e.status, "%s", e.what());
    } catch (const std::exception &e) {
        delete res;
        return fail(OSP_ERR_ALLOC, "%s", e.what());
    }
    *out = (osp_multi_result_t)res;
    return OSP_OK;
}

int osp_spgemm_csc_csr_multi(const int *devices, int ndev, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N, const int64_t *a_colptr,
                             const uint32_t *a_rowidx, const void *a_vals, const int64_t *b_rowptr, const uint32_t *b_colidx,
                             const void *b_vals, const osp_config_t *cfg, osp_multi_context_t *mc_out, osp_multi_result_t *out) {
    if (!mc_out || !out) return fail(OSP_ERR_ARG, "null argument");
    osp_multi_context_t mc = nullptr;
    osp_multi_operands_t ops = nullptr;
    int st = osp_multi_context_create(devices, ndev, &mc);
    if (st) return st;
    st = osp_multi_operands_create(mc, dtype, M, K, N, a_colptr, a_rowidx, a_vals, b_rowptr, b_colidx, b_vals, &ops);
    if (st == OSP_OK) st = osp_spgemm_multi(mc, ops, cfg, out);
    if (ops) osp_multi_operands_destroy(ops);   // the result does not refer to the operands
    if (st) { osp_multi_context_destroy(mc); return st; }
    *mc_out = mc;   // the result's shards live in this context's pools: destroy the result first, then the context
    return OSP_OK;
}

int osp_multi_result_info(osp_multi_result_t r_, osp_multi_info_t *info) {
    MultiResult *r = (MultiResult *)r_;
    if (!r || !info) return fail(OSP_ERR_ARG, "null argument");
    *info = r->info;
    return OSP_OK;
}
int osp_multi_result_shard(osp_multi_result_t r_, int rank, uint64_t *row_begin, uint64_t *row_end, osp_result_t *shard) {
    MultiResult *r = (MultiResult *)r_;
    if (!r || rank < 0 || rank >= (int)r->shard.size()) return fail(OSP_ERR_ARG, "bad result or rank");
    if (row_begin) *row_begin = r->row_bounds[rank];
    if (row_end) *row_end = r->row_bounds[rank + 1];
    if (shard) *shard = (osp_result_t)r->shard[rank];
    return OSP_OK;
}
int osp_multi_result_copy_csr(osp_multi_result_t r_, int64_t *rowptr, uint32_t *colidx, void *vals) {
    MultiResult *r = (MultiResult *)r_;
    if (!r) return fail(OSP_ERR_ARG, "null result");
    const size_t vs = r->dtype == OSP_F32 ? 4 : 8;
    uint64_t base = 0;
    for (size_t g = 0; g < r->shard.size(); g++) {
        Result *sh = r->shard[g];
        const uint64_t r0 = r->row_bounds[g], nr = r->row_bounds[g + 1] - r0, nz = sh->info.nnz_c;
        const int st = osp_result_copy_csr((osp_result_t)sh, rowptr ? rowptr + r0 : nullptr, colidx ? colidx + base : nullptr,
                                           vals ? (char *)vals + base * vs : nullptr, OSP_HOST);
        if (st) return st;
        if (rowptr) for (uint64_t i = 0; i <= nr; i++) rowptr[r0 + i] += (int64_t)base;   // (entry nr is rewritten by the next shard)
        base += nz;
    }
    return OSP_OK;
}
int osp_multi_result_destroy(osp_multi_result_t r) {
    delete (MultiResult *)r;
    return OSP_OK;
}

int osp_result_info(osp_result_t r_, osp_result_info_t *info) {
    Result *r = (Result *)r_;
    if (!r || !info) return fail(OSP_ERR_ARG, "null argument");
    *info = r->info;
    return OSP_OK;
}

int osp_result_copy_csr(osp_result_t r_, int64_t *rowptr, uint32_t *colidx, void *vals, osp_memspace_t space) {
    Result *r = (Result *)r_;
    if (!r) return fail(OSP_ERR_ARG, "null result");
    if (r->partials) return fail(OSP_ERR_ARG, "a result of osp_spgemm_partials holds records, not a CSR: use osp_result_partials");
    OSP_GUARD_BEGIN
    OSP_HIP(hipSetDevice(r->ctx->device));
    hipStream_t s = r->ctx->stream;
    const size_t vs = r->dtype == OSP_F32 ? 4 : 8;
    auto out = [&](void *dst, const void *src, size_t bytes) {
        if (space == OSP_HOST) copy_d2h(dst, src, bytes, s);
        else OSP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    };
    if (rowptr) out(rowptr, r->rowptr, (r->info.M + 1) * sizeof(int64_t));
    if (colidx && r->info.nnz_c) out(colidx, r->colidx, r->info.nnz_c * sizeof(uint32_t));
    if (vals && r->info.nnz_c) out(vals, r->vals, r->info.nnz_c * vs);
    OSP_HIP(hipStreamSynchronize(s));
    return OSP_OK;
    OSP_GUARD_END
}

int osp_result_device_ptrs(osp_result_t r_, const int64_t **rowptr, const uint32_t **colidx, const void **vals) {
    Result *r = (Result *)r_;
    if (!r) return fail(OSP_ERR_ARG, "null result");
    if (r->partials) return fail(OSP_ERR_ARG, "a result of osp_spgemm_partials holds records, not a CSR: use osp_result_partials");
    if (rowptr) *rowptr = r->rowptr;
    if (colidx) *colidx = r->colidx;
    if (vals) *vals = r->vals;
    return OSP_OK;
}

int osp_result_destroy(osp_result_t r_) {
    destroy_result((Result *)r_);
    return OSP_OK;
}

}  // extern "C"
