// osp_pipeline.h -- host orchestration of one product: tile planning, the plan of a panel's long rows (direct / gathered / hub /
// split), the merge of a panel, the panel loop (merge_pipeline) and the producers of partial products.  Part of the ONE
// translation unit osp_api.hip (included there, inside namespace osp, after osp_context.h); split out of it in round 5.
#pragma once

// run `stmt` with RA = the context's ranking variant as a compile-time constant
#define OSP_WITH_RA(ctx_, ...)                                               \
    do {                                                                     \
        if ((ctx_)->rank_atomic) { constexpr bool RA = true; __VA_ARGS__; }  \
        else { constexpr bool RA = false; __VA_ARGS__; }                     \
    } while (0)

// Where the partial products of a panel come from.
template <class T> struct Producer {
    virtual ~Producer() {}
    // enqueue kernels that fill stage[0 .. row_off[r1]-row_off[r0]) for rows [r0,r1)
    // (cells / qstage: the plan and the second buffer of the panel's direct rows, osp_kernels.h store_direct; null
    // when the panel has none)
    // (hub: cells and run table of the panel's hub rows, osp_kernels.h "HUB rows"; null when the panel has none)
    // (compact: the panel has gathered rows -- chunks nobody writes; osp_kernels.h, multiply_kernel IND;
    //  has_long: it has rows longer than a tile)
    virtual void produce(uint64_t r0, uint64_t r1, bool whole, uint64_t base, uint64_t count,
                         Part<T> *stage, PhaseTimer &tm, const uint32_t *cells = nullptr, Part<T> *qstage = nullptr,
                         const HubArgs *hub = nullptr, bool compact = false, bool has_long = true, bool desc_only = false, bool walk_all = false) = 0;
};

// ---- rows of partial products -> merged rows -------------------------------------------------------
template <class T> struct MergeIO {
    Part<T> *stage;                      // partial products of rows [r0,r1), addressed row_off[r] - base
    const uint64_t *row_off; uint64_t r0, r1, base;
    int64_t *c_rowptr; uint32_t *c_col; T *c_val;  // output (c_rowptr indexed by absolute row id)
    const uint64_t *out_in; uint64_t *out_out;    // entries written before / after this call (device)
    ChunkTable<T> ct{};                           // row-wise variant: rows that fit a tile are computed in the tile kernel
    uint32_t *abort_word = nullptr;               // set by a look-back that gave up (merge_tiles_kernel): the product is an error
    // gathered short rows (osp_kernels.h, GatherArgs): their run table, the first run of every row, B (null: they are staged)
    const RunDesc<T> *runs0 = nullptr; const uint32_t *rowfirst0 = nullptr; const uint32_t *b_colidx = nullptr; const T *b_vals = nullptr;
};
template <class T> struct ShortRuns {
    const RunDesc<T> *runs0 = nullptr; const uint32_t *rowfirst0 = nullptr; const uint32_t *b_colidx = nullptr; const T *b_vals = nullptr;
};

struct TilePlan {
    uint32_t *tile_rows = nullptr;  // first row of every tile
    uint32_t ntiles = 0;
    uint32_t *long_rows = nullptr;  // rows with more partial products than one tile
    uint32_t nlong = 0;
};

// Greedy tile packing (coarse blocks of ~8 tiles, one walker thread per block) + the list of long rows.
// total: entries of rows [r0, r1); nforce: upper bound of the rows flagged in force_start -- what bounds the number of
// coarse blocks on the host (the count itself is only read by the kernels).
static TilePlan plan_tiles(Context *ctx, Scratch &sc, const uint64_t *row_off, uint64_t r0, uint64_t r1, uint64_t base,
                           uint32_t cap, uint32_t max_rows, const uint8_t *force_start, uint64_t total, uint64_t nforce) {
    hipStream_t s = ctx->stream;
    const uint64_t nr = r1 - r0;
    TilePlan pl;
    uint32_t *flag_scan = sc.get<uint32_t>(nr + 1);
    uint32_t *tmp_rows = sc.get<uint32_t>(nr + 1);
    uint64_t *scan_tmp = sc.get<uint64_t>(scan_scratch_entries(std::max<uint64_t>(nr + 1, 16)));
    const uint64_t slot = 8ull * cap;
    CoarseStartFlag csf{row_off, r0, base, slot, force_start};
    device_exclusive_scan<CoarseStartFlag, uint32_t>(csf, nr, flag_scan, (uint32_t *)scan_tmp, s);
    compact_flagged_kernel<CoarseStartFlag><<<grid_for(nr, 256), 256, 0, s>>>(csf, flag_scan, nr, r0, tmp_rows);
    // flagged: row 0, the forced rows, one row per slot boundary crossed, every 65536th row
    const uint32_t ncb = (uint32_t)std::min<uint64_t>(nr, 2 + nforce + total / slot + nr / 65536);
    const uint32_t *ncb_p = flag_scan + nr;
    uint32_t *cb_cnt = sc.get<uint32_t>((uint64_t)ncb + 1);
    tile_walk_kernel<<<grid_for(ncb, 128), 128, 0, s>>>(tmp_rows, ncb_p, ncb, r1, row_off, cap, max_rows, nullptr, cb_cnt, nullptr);
    device_exclusive_scan<LoadU32, uint32_t>(LoadU32{cb_cnt}, ncb, cb_cnt, (uint32_t *)scan_tmp, s);
    // the long rows' list needs nothing of the tile list: both counts come home with one wait
    HeavyRowFlag hrf{row_off, r0, cap};
    uint32_t *long_scan = sc.get<uint32_t>(nr + 1);
    uint32_t *long_tmp = sc.get<uint32_t>(scan_scratch_entries(std::max<uint64_t>(nr + 1, 16)));
    device_exclusive_scan<HeavyRowFlag, uint32_t>(hrf, nr, long_scan, long_tmp, s);
    pl.long_rows = sc.get<uint32_t>(nr + 1);
    compact_flagged_kernel<HeavyRowFlag><<<grid_for(nr, 256), 256, 0, s>>>(hrf, long_scan, nr, r0, pl.long_rows);
    { Gather g(s); g.add(&pl.ntiles, (const uint32_t *)cb_cnt + ncb); g.add(&pl.nlong, (const uint32_t *)long_scan + nr); g.wait(); }
    pl.tile_rows = sc.get<uint32_t>((uint64_t)pl.ntiles + 1);
    tile_walk_kernel<<<grid_for(ncb, 128), 128, 0, s>>>(tmp_rows, ncb_p, ncb, r1, row_off, cap, max_rows, cb_cnt, nullptr, pl.tile_rows);
    return pl;
}

// What the planner of direct rows reads -- the chunk table of the symbolic phase: the chunks (non-zeros of A) in (row, k)
// order with their staging offsets and B rows -- and the chunk offsets it replaces by descriptors (osp_kernels.h,
// store_direct).  direct_max: longest row (partial products) that is planned as a direct row.
struct DirectSrc {
    const uint32_t *rowfirst; const uint64_t *off; const uint32_t *bs; const uint32_t *perm; const uint32_t *b_colidx;
    uint64_t *chunk_off;
    uint64_t direct_max;
    // hub rows (osp_split.h, hub_plan_kernel): B's pointer array for the run table, which is made when the first panel with hub
    // rows asks for it and lives in `keep` as long as the product
    const int64_t *b_rowptr = nullptr;
    uint64_t K = 0, nnz_b = 0;
    Scratch *keep = nullptr;
    mutable HubTables hub{};
    mutable bool hub_refused = false;   // a panel's hub rows turned out to be runs of a few records each: the product keeps the stretch split
    // gathered rows (osp_kernels.h): direct rows without an over-long range are not written by the multiply at all; the tile
    // kernel forms their records from run descriptors.  a_vals: indexed by `perm`; gstat: rows / partial products / runs (device)
    bool gather = false;
    bool expand_rows = false;   // the short rows are gathered too: plainly staged long rows are expanded row by row (expand_rows_kernel)
    // ... and then nobody reads the chunk offsets unless a panel has rows written through cells (hub rows, the fallbacks of
    // the gathered rows): they are made when the first such panel is planned (ensure_chunk_off), not by every product
    // (sym_scatter_offsets_kernel: 1 ms of scattered 8-byte stores on the headline, 0.08 of the web-Google shape's 2 ms)
    mutable bool chunk_off_ready = true;
    const uint32_t *rows_sorted = nullptr;
    const uint64_t *row_off = nullptr;
    uint64_t rw_cap = 0, nnz = 0;
    void ensure_chunk_off(hipStream_t s) const {
        if (chunk_off_ready) return;
        sym_scatter_offsets_kernel<<<grid_for(nnz, 256), 256, 0, s>>>(perm, off, rows_sorted, row_off, rw_cap, nnz, chunk_off);
        chunk_off_ready = true;
    }
    const void *a_vals = nullptr, *b_vals = nullptr;   // a_vals: indexed by `perm`, or (av_in_order) the values in (row, k) order
    bool av_in_order = false;
    unsigned long long *gstat = nullptr;
};

// What is decided about a panel BEFORE its partial products exist (plan_panel) and used after the multiply (merge_panel).
template <class T> struct PanelPlan {
    Scratch sc;           // owns every array below; released when the panel is done
    TilePlan p0;          // level-0 tiles and the list of long rows
    uint32_t max_rows = 0;
    // ---- long rows ----
    uint64_t *hoff = nullptr, *hscan_tmp = nullptr, *blkbase = nullptr, *hbase = nullptr, *vbase = nullptr, *cellbase = nullptr;
    uint8_t *hbits = nullptr, *hmode = nullptr;
    uint32_t *nstretch = nullptr;
    uint64_t nh = 0, nblocks = 0, nvirt = 0, ncell = 0;
    uint64_t mode_rows[3] = {0, 0, 0}, mode_partials[3] = {0, 0, 0};   // per kMode*: how many long rows, how many partial products
    uint32_t *ghist = nullptr, *ghist_tmp = nullptr;
    Part<T> *qstage = nullptr;        // the second buffer: long rows by column range
    uint64_t *vrow_off = nullptr;     // its segments ("virtual rows"): offsets, first-of-row flags, column bounds
    uint8_t *vfirst = nullptr;
    uint32_t *vcol0 = nullptr, *vcol1 = nullptr;
    uint32_t *cells = nullptr;        // direct rows: range tables and (chunk, range) cells
    HubArgs hub{};                    // hub rows: (chunk, run) cells and B's run table; cells == nullptr: the panel has none
    GatherArgs<T> ga{};               // gathered rows: the run table (runs == nullptr: the panel has none) ...
    uint32_t *vrun_off = nullptr, *vrun_end = nullptr;   // ... and every segment's descriptors in it
    uint32_t *nwritten = nullptr;     // direct rows that are not gathered (device; 0: only the hub rows need the column-major multiply)
    uint64_t *xjobbase = nullptr;     // expand_rows_kernel's jobs: first job of every long row (null: the multiply stages the rows)
    uint64_t xjobs_bound = 0, xpartials = 0;
    bool expand_ok = false;           // the panel's plainly staged long rows (if any) have jobs: the column-major multiply need not stage them
    bool may_write = false;           // some planned rows of the panel may have been written through cells (OSP_GATHER_OVER=0)
    explicit PanelPlan(Context *c) : sc(c) {}
};

// totals[mode] += rows, totals[3 + mode] += partial products, per mode of the long rows
__global__ void mode_totals_kernel(const uint32_t *rows, uint32_t nlong, const uint64_t *row_off, const uint8_t *hmode,
                                   unsigned long long *totals) {
    // (grid-stride, few workgroups: every wave ends in up to six atomics on six hot words -- one wave per 64 rows made them
    // 0.18 ms per panel on R-MAT-22)
    uint64_t part[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nlong; h += gridDim.x * blockDim.x) {
        const uint64_t U = row_off[rows[h] + 1] - row_off[rows[h]];
        const uint8_t m = hmode[h];
#pragma unroll
        for (int mode = 0; mode < 3; mode++) { part[mode] += m == mode ? U : 0ull; cnt[mode] += m == mode ? 1ull : 0ull; }
    }
    for (uint8_t mode = 0; mode < 3; mode++) {
        const uint64_t p = wave_reduce_sum<uint64_t>(part[mode]), c = wave_reduce_sum<uint64_t>(cnt[mode]);
        if (lane_id() == 0 && c) { atomicAdd(&totals[mode], (unsigned long long)c); atomicAdd(&totals[3 + mode], (unsigned long long)p); }
    }
}

// Before the multiply: level-0 tiles, the long rows and how each of them will reach the tile kernel, the second buffer
// and its segment tables -- and, for direct rows, the plan the multiply writes them by.
template <class T>
static void plan_panel(Context *ctx, Result *res, PhaseTimer &tm, PanelPlan<T> &pl, const uint64_t *row_off, uint64_t r0, uint64_t r1,
                       uint64_t base, uint64_t count, int colbits, const DirectSrc *ds) {
    hipStream_t s = ctx->stream;
    Scratch &sc = pl.sc;
    constexpr uint32_t kCap = (uint32_t)TileCap<T>::value;
    pl.max_rows = (uint32_t)std::min<uint64_t>(kTileMaxRows, colbits >= 32 ? 1ull : (1ull << (32 - colbits)));
    pl.p0 = plan_tiles(ctx, sc, row_off, r0, r1, base, kCap, pl.max_rows, nullptr, count, 0);
    res->info.light_tiles += pl.p0.ntiles - pl.p0.nlong;
    if (!pl.p0.nlong) return;
    const uint32_t nlong = pl.p0.nlong;
    pl.hoff = sc.get<uint64_t>((uint64_t)nlong + 1);
    pl.hscan_tmp = sc.get<uint64_t>(scan_scratch_entries(nlong));
    device_exclusive_scan<HeavyLen, uint64_t>(HeavyLen{pl.p0.long_rows, row_off}, nlong, pl.hoff, pl.hscan_tmp, s);
    pl.hbits = sc.get<uint8_t>(nlong);
    pl.hmode = sc.get<uint8_t>(nlong);
    pl.nstretch = sc.get<uint32_t>(nlong);
    uint32_t *nseg = sc.get<uint32_t>(nlong);
    uint64_t *nhist = sc.get<uint64_t>(nlong), *ncellh = sc.get<uint64_t>(nlong);
    const bool want_gather = ds && ds->gather;
    uint64_t *nrund = want_gather ? sc.get<uint64_t>(nlong) : nullptr, *rdbase = want_gather ? sc.get<uint64_t>((uint64_t)nlong + 1) : nullptr;
    uint64_t nrd = 0;
    pl.blkbase = sc.get<uint64_t>((uint64_t)nlong + 1); pl.hbase = sc.get<uint64_t>((uint64_t)nlong + 1);
    pl.vbase = sc.get<uint64_t>((uint64_t)nlong + 1); pl.cellbase = sc.get<uint64_t>((uint64_t)nlong + 1);
    unsigned long long *totals = (unsigned long long *)sc.get<uint64_t>(6);
    // debugging aid: OSP_SPLIT_ROW_MAX moves the boundary between the two split kernels (tests run both on small inputs)
    const uint64_t row_max = getenv("OSP_SPLIT_ROW_MAX") ? strtoull(getenv("OSP_SPLIT_ROW_MAX"), nullptr, 10) : kSplitRowMax;
    // no more ranges than make a range as narrow as the dense accumulators take (osp_split.h, kDenseBits): beyond that
    // a finer split only shortens the runs the scatter writes -- whatever a range of <= 2048 columns holds is summed
    // without a sort.  It bites for rows with more than one product per 8 columns of B (kSplitTarget = 256 per 2048 columns).  (Until late in round 3 the cap
    // was never below kSplitRowBits, i.e. without effect for N < 2^20: a product with dense output rows -- 32768^2, 634
    // entries per row -- sorted 512 ranges of 64 columns per row, 370 ms; with 16 ranges of 2048 columns it takes 197.)
    const int bits_cap = std::max(colbits - kDenseBits, 1);
    // Hub rows: with a chunk table at hand the rows beyond the one-workgroup planner are written by the multiply as well, into
    // 2^hub_b uniform column blocks (no narrower than a dense accumulator, no more than the stretch split's 4096), instead of
    // being moved by the stretch split afterwards.  It needs the lane order of LDS atomics (the context's self-test) and B's
    // run table; OSP_HUB=0 keeps the stretch split (debugging aid, A/B timing).
    const int hub_b_want = std::min(kSplitMaxBits, bits_cap);
    const bool hub_env = !(getenv("OSP_HUB") && atoi(getenv("OSP_HUB")) == 0);
    int hub_b = 0;
    uint64_t ndcell = 0, tot[6] = {0, 0, 0, 0, 0, 0};
    // ... and it pays only where such rows hold a good part of the panel's products: the multiply of a panel with hub rows is
    // the instantiation that knows their descriptors (74 registers instead of 68) for ALL its products, and the plan has fixed
    // costs.  Measured (round 4): Graph500 scale 22 streamed 3.48 -> 3.23 s, scale 20 387 -> 368 ms with them (stretch rows:
    // more than half of the products); R-MAT-22 "mild" (2 %) 236 -> 250 ms.  OSP_HUB_MIN_SHARE moves the threshold.
    // fine bins of the direct rows' planner: 2^direct_fine per kSplitTarget products (osp_split.h, split_params_kernel)
    const int direct_fine = 2;   // (0 = the bins of round 3; measured in round 4, MEASUREMENTS 1.x: the ranges fill their tiles to 97 % instead of 92)
    const double hub_min_share = getenv("OSP_HUB_MIN_SHARE") ? atof(getenv("OSP_HUB_MIN_SHARE")) : 0.2;
    bool hub_decided = false;
    for (int attempt = 0; attempt < 3; attempt++) {
        const bool hub_possible = ds && ds->b_rowptr && ds->keep && ctx->rank_atomic && hub_env && !ds->hub_refused;
        if (!hub_possible) hub_b = 0;
        split_params_kernel<<<grid_for(nlong, 256), 256, 0, s>>>(pl.p0.long_rows, nlong, row_off, colbits, row_max, bits_cap, ds ? ds->rowfirst : nullptr,
                                                                 ds ? ds->direct_max : 0ull, kCap, pl.hbits, pl.hmode, pl.nstretch, nseg, nhist, ncellh,
                                                                 hub_b, direct_fine, ds ? nrund : nullptr);
        zero_async(s, {{totals, 6 * sizeof(uint64_t)}});
        mode_totals_kernel<<<std::min(grid_for(nlong, 256), 64u), 256, 0, s>>>(pl.p0.long_rows, nlong, row_off, pl.hmode, totals);
        device_exclusive_scan<LoadU32As64, uint64_t>(LoadU32As64{pl.nstretch}, nlong, pl.blkbase, pl.hscan_tmp, s);
        device_exclusive_scan<LoadU32As64, uint64_t>(LoadU32As64{nseg}, nlong, pl.vbase, pl.hscan_tmp, s);
        device_exclusive_scan<LoadU64, uint64_t>(LoadU64{nhist}, nlong, pl.hbase, pl.hscan_tmp, s);
        if (ds) device_exclusive_scan<LoadU64, uint64_t>(LoadU64{ncellh}, nlong, pl.cellbase, pl.hscan_tmp, s);
        if (ds && nrund) device_exclusive_scan<LoadU64, uint64_t>(LoadU64{nrund}, nlong, rdbase, pl.hscan_tmp, s);
        ndcell = 0;
        nrd = 0;
        {
            Gather g(s);
            g.add(&pl.nh, (const uint64_t *)pl.hoff + nlong); g.add(&pl.nblocks, (const uint64_t *)pl.blkbase + nlong);
            g.add(&pl.nvirt, (const uint64_t *)pl.vbase + nlong); g.add(&pl.ncell, (const uint64_t *)pl.hbase + nlong);
            if (ds) g.add(&ndcell, (const uint64_t *)pl.cellbase + nlong);
            if (ds && nrund) g.add(&nrd, (const uint64_t *)rdbase + nlong);
            for (int i = 0; i < 6; i++) g.add(&tot[i], (const uint64_t *)totals + i);
            g.wait();
        }
        // the cells of a panel are addressed with 32 bits (and are device memory beside the staging buffers): a panel whose
        // plan would not fit splits its long rows after the multiply instead
        if (ds && ndcell >= 0xffffffffull) { ds = nullptr; hub_b = 0; continue; }
        // first look at the panel: do its stretch rows hold enough of it to be planned as hub rows?  (once more, with their blocks)
        if (hub_possible && !hub_decided && hub_b == 0 && pl.nblocks && (double)tot[3 + kModeStretch] >= hub_min_share * (double)count) {
            hub_decided = true;
            hub_b = hub_b_want;
            continue;
        }
        break;
    }
    for (int m = 0; m < 3; m++) { pl.mode_rows[m] = tot[m]; pl.mode_partials[m] = tot[3 + m]; }
    if (pl.nh >= 0xffffffffull) throw Error(OSP_ERR_CAPACITY, "long rows of one panel exceed 2^32 partial products");
    res->info.heavy_rows += nlong;
    res->info.heavy_partials += pl.nh;
    res->info.direct_rows += pl.mode_rows[kModeDirect];
    res->info.direct_partials += pl.mode_partials[kModeDirect];
    if (pl.ncell >= 0xffffffffull || pl.nblocks >= 0x7fffffffull || pl.nvirt >= 0xffffffffull || ndcell >= 0xffffffffull)
        throw Error(OSP_ERR_CAPACITY, "split histogram too large");
    if (getenv("OSP_VERBOSE"))
        fprintf(stderr, "[osp]   panel rows [%llu,%llu): %u tiles, %u long rows with %llu partial products -> %llu segments; direct: %llu rows, %llu "
                        "partial products, %llu cells\n",
                (unsigned long long)r0, (unsigned long long)r1, pl.p0.ntiles, nlong, (unsigned long long)pl.nh, (unsigned long long)pl.nvirt,
                (unsigned long long)pl.mode_rows[kModeDirect], (unsigned long long)pl.mode_partials[kModeDirect], (unsigned long long)ndcell);
    pl.ghist = sc.get<uint32_t>(pl.ncell + 1);
    pl.ghist_tmp = sc.get<uint32_t>(scan_scratch_entries(pl.ncell + 1));
    pl.vrow_off = sc.get<uint64_t>(pl.nvirt + 1);
    pl.vfirst = sc.get<uint8_t>(pl.nvirt + 1);
    pl.vcol0 = sc.get<uint32_t>(pl.nvirt + 1);
    pl.vcol1 = sc.get<uint32_t>(pl.nvirt + 1);
    if (pl.mode_rows[kModeDirect]) {
        pl.cells = sc.get<uint32_t>(ndcell);
        GatherPlan gp{};
        RunDesc<T> *runs = nullptr;
        // (the run table is addressed with 32 bits; a panel whose bound does not fit writes its direct rows as before)
        // (debugging aid: OSP_GATHER_MAX_RUNS lowers that limit, so that a test reaches the fallback)
        static const uint64_t max_runs = getenv("OSP_GATHER_MAX_RUNS") ? strtoull(getenv("OSP_GATHER_MAX_RUNS"), nullptr, 10) : 0xffffffffull;
        if (ds->gather && nrund && nrd < max_runs) {
            runs = sc.get<RunDesc<T>>(std::max<uint64_t>(nrd, 1));
            pl.vrun_off = sc.get<uint32_t>(pl.nvirt + 1);
            pl.vrun_end = sc.get<uint32_t>(pl.nvirt + 1);
            gp.rdbase = rdbase; gp.vrun_off = pl.vrun_off; gp.vrun_end = pl.vrun_end;
            gp.rowruns = sc.get<uint32_t>(nlong);
            pl.nwritten = gp.nwritten = sc.get<uint32_t>(1);
            // kNoRuns: segments / rows that are not gathered (ONE kernel: a hipMemsetAsync costs the host tens of microseconds)
            fill_async(s, {{pl.vrun_off, (pl.nvirt + 1) * sizeof(uint32_t), kNoRuns}, {gp.rowruns, (uint64_t)nlong * sizeof(uint32_t), kNoRuns},
                           {pl.nwritten, sizeof(uint32_t), 0u}});
            gp.over = !(getenv("OSP_GATHER_OVER") && atoi(getenv("OSP_GATHER_OVER")) == 0);
            gp.av_in_order = ds->av_in_order ? 1u : 0u;
            pl.may_write = !gp.over;
            pl.ga.runs = runs; pl.ga.b_colidx = ds->b_colidx; pl.ga.b_vals = (const T *)ds->b_vals;
        }
        // rows the multiply will write -- direct rows that are not gathered, hub rows -- get descriptors in the chunk offsets, and
        // the gathered rows beside them their skip marks: the offsets must exist first
        if (!(runs && gp.over) || (hub_b && pl.nblocks)) ds->ensure_chunk_off(s);
        gp.mark_skipped = ds->chunk_off_ready ? 1u : 0u;
        tm.begin(PH_PLAN_K, s);
        if (nlong <= 4u * ctx->cus)   // few rows: the launch lasts as long as its longest row (osp_split.h, kDirectCellsBig)
            direct_plan_kernel<T, kDirectCellsBig><<<nlong, kDirectThreads, 0, s>>>(pl.p0.long_rows, nlong, pl.hmode, pl.hbits, nseg, pl.vbase, pl.hoff,
                                                                                 pl.cellbase, row_off, colbits, kCap, ds->rowfirst, ds->off, ds->bs, ds->perm,
                                                                                 ds->b_colidx, pl.vrow_off, pl.vcol0, pl.vcol1, pl.cells, ds->chunk_off, gp,
                                                                                 (const T *)ds->a_vals, runs);
        else
        direct_plan_kernel<T><<<nlong, kDirectThreads, 0, s>>>(pl.p0.long_rows, nlong, pl.hmode, pl.hbits, nseg, pl.vbase, pl.hoff, pl.cellbase, row_off,
                                                              colbits, kCap, ds->rowfirst, ds->off, ds->bs, ds->perm, ds->b_colidx, pl.vrow_off,
                                                              pl.vcol0, pl.vcol1, pl.cells, ds->chunk_off, gp, (const T *)ds->a_vals, runs);
        tm.end(PH_PLAN_K);
        if (gp.rowruns && ds->gstat) gather_stats_kernel<<<std::min(grid_for(nlong, 256), 64u), 256, 0, s>>>(pl.p0.long_rows, nlong, row_off, gp.rowruns, ds->gstat);
#ifdef OSP_PLAN_PROF
        if (getenv("OSP_VERBOSE")) {
            unsigned long long hp[8] = {0}, z[8] = {0};
            OSP_HIP(hipStreamSynchronize(s));
            (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(osp_plan_prof), sizeof(hp));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(osp_plan_prof), z, sizeof(z));
            double tot = 0;
            for (int k = 0; k < 6; k++) tot += (double)hp[k];
            fprintf(stderr, "[osp]   planner cycles: header %.1f %%, chunk descriptors %.1f %%, histogram pass %.1f %%, grouping %.1f %%, cell pass %.1f %%, "
                            "prefixes + output %.1f %%; %.0f cycles per row, %u rows\n", 100 * hp[0] / tot, 100 * hp[1] / tot, 100 * hp[2] / tot, 100 * hp[3] / tot,
                    100 * hp[4] / tot, 100 * hp[5] / tot, tot / std::max(1u, nlong), nlong);
        }
#endif
        res->info.direct_plan_launches++;
        dbg_sync(s, "plan of the direct rows");
    }
    if (hub_b && pl.nblocks) {
        // ---- hub rows: B's run table (once per product), per-job block totals, ONE scan for the segment offsets, cells ----
        if (!ds->hub.sx) {
            Scratch &keep = *ds->keep;
            const uint64_t nb = ds->nnz_b;
            Scratch tmp(ctx);
            uint8_t *rowstart = tmp.get<uint8_t>(nb + 4);
            uint32_t *sx = keep.get<uint32_t>(nb + 1), *runstart = keep.get<uint32_t>(nb + 1);
            uint16_t *runblk = keep.get<uint16_t>(nb + 2);
            uint32_t *scan_tmp = tmp.get<uint32_t>(scan_scratch_entries(nb + 1));
            OSP_HIP(hipMemsetAsync(rowstart, 0, nb + 4, s));
            hub_rowstart_kernel<<<grid_for(ds->K, 256), 256, 0, s>>>(ds->b_rowptr, ds->K, rowstart);
            const int sh = colbits - hub_b;
            const HubRunFlag flag{rowstart, ds->b_colidx, sh};
            device_exclusive_scan<HubRunFlag, uint32_t>(flag, nb, sx, scan_tmp, s);
            compact_flagged_kernel<HubRunFlag><<<grid_for(nb, 256), 256, 0, s>>>(flag, sx, nb, 0, runstart);
            hub_runstart_end_kernel<<<1, 1, 0, s>>>(sx + nb, runstart, (uint32_t)nb);
            OSP_HIP(hipMemsetAsync(runblk, 0, (nb + 2) * sizeof(uint16_t), s));
            hub_runblk_kernel<<<std::min(grid_for(nb, 256), 4096u), 256, 0, s>>>(sx + nb, runstart, ds->b_colidx, sh, runblk);
            OSP_HIP(hipStreamSynchronize(s));   // (tmp goes back to the pool; everything that read it is done)
            ds->hub.sx = sx; ds->hub.runstart = runstart; ds->hub.runblk = runblk; ds->hub.sh = sh;
        }
        tm.begin(PH_HUB_K, s);
        uint64_t *jobruns = sc.get<uint64_t>(pl.nblocks + 1);
        uint64_t *jobscan_tmp = sc.get<uint64_t>(scan_scratch_entries(pl.nblocks + 1));
        const size_t hub_lds = sizeof(uint32_t) << hub_b;
        hub_plan_kernel<false><<<(unsigned)pl.nblocks, kHubThreads, hub_lds, s>>>(pl.p0.long_rows, nlong, pl.blkbase, pl.hbase, pl.hbits, pl.nstretch, row_off,
                                                                           ds->rowfirst, ds->off, ds->bs, ds->perm, ds->b_colidx, ds->hub, pl.ghist,
                                                                           pl.hoff, jobruns, nullptr, nullptr, nullptr);
        device_exclusive_scan<LoadU32, uint32_t>(LoadU32{pl.ghist}, pl.ncell, pl.ghist, pl.ghist_tmp, s);
        device_exclusive_scan<LoadU64, uint64_t>(LoadU64{jobruns}, pl.nblocks, jobruns, jobscan_tmp, s);
        const uint64_t ncells = d2h(jobruns + pl.nblocks, s);
        if (ncells >= (1ull << 40)) throw Error(OSP_ERR_CAPACITY, "hub rows: too many runs in one panel");
        // Runs of a few records each are not worth writing one by one: hub rows where a run holds OSP_HUB_MIN_RUN records on
        // average (default 4).  Measured with the run-block table and plain stores for hub records (round 4, one box, hub rows
        // on / off): R-MAT-19 "mild" at edge factor 64, 4.2 records per run, 293 / 326 ms; Graph500 scale 22, 4.7-6.9 records per
        // run, 13 % faster; scale 16-20, 7.1-43 records per run, 6-18 % faster; 2.4 and 1.4 records per run (R-MAT-20 / 22 "mild",
        // a tenth and a twentieth of the products in such rows): no difference either way.  (Until the run-block table the 4.2 case
        // lost -- 324-354 against 315 ms -- and the threshold was six, or 4.5 where the stretch split's rounds were shorter still.)
        // Otherwise the panel -- and the rest of the product -- keeps the stretch split (with the blocks already chosen: its
        // histogram has the same layout).
        if (getenv("OSP_VERBOSE"))
            fprintf(stderr, "[osp]   hub rows: %llu rows, %llu products in %llu runs (%.2f records per run), %llu jobs, 2^%d blocks\n",
                    (unsigned long long)pl.mode_rows[kModeStretch], (unsigned long long)pl.mode_partials[kModeStretch], (unsigned long long)ncells,
                    ncells ? (double)pl.mode_partials[kModeStretch] / (double)ncells : 0.0, (unsigned long long)pl.nblocks, hub_b);
        const double min_run = getenv("OSP_HUB_MIN_RUN") ? atof(getenv("OSP_HUB_MIN_RUN")) : 4.0;
        const double run = ncells ? (double)pl.mode_partials[kModeStretch] / (double)ncells : 0.0;
        const bool accept = run >= min_run;
        if (!accept) {
            ds->hub_refused = true;
            tm.end(PH_HUB_K);
        } else {
            uint32_t *hcells = sc.get<uint32_t>(ncells);
            ds->ensure_chunk_off(s);
            hub_plan_kernel<true><<<(unsigned)pl.nblocks, kHubThreads, hub_lds, s>>>(pl.p0.long_rows, nlong, pl.blkbase, pl.hbase, pl.hbits, pl.nstretch, row_off,
                                                                              ds->rowfirst, ds->off, ds->bs, ds->perm, ds->b_colidx, ds->hub, pl.ghist,
                                                                              pl.hoff, nullptr, jobruns, hcells, ds->chunk_off);
            tm.end(PH_HUB_K);
            pl.hub.cells = hcells;
            pl.hub.sx = ds->hub.sx;
            res->info.hub_plan_launches++;
            res->info.hub_cells += ncells;
            res->info.hub_rows += pl.mode_rows[kModeStretch];
            res->info.hub_partials += pl.mode_partials[kModeStretch];
        }
        dbg_sync(s, "plan of the hub rows");
    }
    // With the short rows gathered, the long rows that are staged plainly -- split rows, stretch rows unless they are hub rows -- are
    // expanded row by row (expand_rows_kernel) instead of column by column: their jobs.
    if (ds && ds->expand_rows) {
        pl.expand_ok = true;
        const bool hubs = pl.hub.cells != nullptr;
        const uint64_t rows_x = pl.mode_rows[kModeSplitRow] + (hubs ? 0 : pl.mode_rows[kModeStretch]);
        pl.xpartials = pl.mode_partials[kModeSplitRow] + (hubs ? 0 : pl.mode_partials[kModeStretch]);
        if (rows_x) {
            pl.xjobbase = sc.get<uint64_t>((uint64_t)nlong + 1);
            device_exclusive_scan<ExpandJobs, uint64_t>(ExpandJobs{pl.p0.long_rows, row_off, pl.hmode, kModeSplitRow, hubs ? kModeSplitRow : kModeStretch},
                                                        nlong, pl.xjobbase, pl.hscan_tmp, s);
            pl.xjobs_bound = pl.xpartials / kExpandJob + rows_x;
        }
    }
}

// One panel after the multiply: long rows that are not direct are split into column-range segments; the tiles of all
// segments take their long row's place in ONE offset chain, so every merged entry is written once, straight to the
// final CSR.
template <class T>
static void merge_panel(Context *ctx, Result *res, PhaseTimer &tm, const MergeIO<T> &io, int colbits, PanelPlan<T> &pl) {
    hipStream_t s = ctx->stream;
    Scratch &sc = pl.sc;
    constexpr uint32_t kCap = (uint32_t)TileCap<T>::value;
    const uint32_t max_rows = pl.max_rows;
#ifdef OSP_CHECK_DESC
    crumbs_init();
#endif
    const uint64_t r1 = io.r1, base = io.base;
    const TilePlan &p0 = pl.p0;

    MergeLevels<T> lv{};
    lv.stage[0] = io.stage; lv.row_off[0] = io.row_off; lv.base[0] = base; lv.c_rowptr[0] = io.c_rowptr; lv.heavy_nnz[0] = nullptr;
    uint32_t ntot = p0.ntiles;
    TileDesc *desc = nullptr;
    // level-1 state (long rows by column range)
    TilePlan p1;
    uint64_t *vbase = pl.vbase;
    int64_t *vptr = nullptr;
    uint64_t *seg_src = nullptr;   // too-long segments: where their reduced entries sit in the second buffer
    uint32_t *seg_nnz = nullptr;
    Part<T> *qstage = pl.qstage;
    if (p0.nlong) {
        const uint32_t nlong = p0.nlong;
        uint64_t *hoff = pl.hoff, *hscan_tmp = pl.hscan_tmp, *blkbase = pl.blkbase, *hbase = pl.hbase;
        uint8_t *hbits = pl.hbits;
        uint32_t *nstretch = pl.nstretch, *ghist = pl.ghist, *ghist_tmp = pl.ghist_tmp;
        const uint64_t nh = pl.nh, nblocks = pl.nblocks, nvirt = pl.nvirt, ncell = pl.ncell;
        uint64_t *vrow_off = pl.vrow_off;
        uint8_t *vfirst = pl.vfirst;
        // ---- rows that are not direct: one stable split by column range into the second buffer ----
        // rows up to kSplitRowMax: one workgroup each (histogram, scan and scatter in one kernel)
        if (pl.mode_rows[kModeSplitRow]) {
            tm.begin(PH_SPLIT_K);
            OSP_WITH_RA(ctx, split_row_kernel<T, RA><<<nlong, kSplitRowThreads, 0, s>>>(p0.long_rows, nlong, hbits, pl.hmode, vbase, hoff, io.row_off,
                                                                                     base, colbits, io.stage, qstage, vrow_off));
            tm.end(PH_SPLIT_K);
            res->info.split_launches++;
            res->info.split_partials += pl.mode_partials[kModeSplitRow];
            dbg_sync(s, "split: one-workgroup rows");
        }
        if (nblocks && !pl.hub.cells) {  // longer rows: one workgroup per 4096-entry stretch, offsets from a device-wide scan
            split_count_kernel<<<(unsigned)nblocks, kSplitThreads, 0, s>>>(p0.long_rows, nlong, blkbase, hbase, hbits, nstretch, io.row_off,
                                                                         base, colbits, (const char *)io.stage, (uint32_t)sizeof(Part<T>), ghist);
            device_exclusive_scan<LoadU32, uint32_t>(LoadU32{ghist}, ncell, ghist, ghist_tmp, s);
            OSP_WITH_RA(ctx, split_scatter_kernel<T, RA><<<(unsigned)nblocks, kSplitThreads, 0, s>>>(
                                 p0.long_rows, nlong, blkbase, hbase, hbits, nstretch, io.row_off, base, colbits, io.stage, ghist, hoff, qstage));
        }
        dbg_sync(s, "split: stretch rows");
        split_vrows_kernel<<<grid_for(nvirt + 1, 256), 256, 0, s>>>(nlong, vbase, hbase, nstretch, hbits, pl.hmode, colbits, ghist, hoff, nvirt, nh,
                                                                   vrow_off, vfirst, pl.vcol0, pl.vcol1);
        OSP_HIP(hipGetLastError());
        dbg_sync(s, "split: segment offsets");
        // ---- tiles over the segments; a tile never spans two long rows ----
        p1 = plan_tiles(ctx, sc, vrow_off, 0, nvirt, 0, kCap, max_rows, vfirst, nh, nlong);
        dbg_sync(s, "tiles over the segments");
        vptr = (int64_t *)sc.get<uint64_t>(nvirt + 1);
        lv.stage[1] = qstage; lv.row_off[1] = vrow_off; lv.base[1] = 0; lv.c_rowptr[1] = vptr;
        if (p1.nlong) {
            // ---- segments that are still too long ----
            const uint32_t nseg_long = p1.nlong;
            seg_src = sc.get<uint64_t>(nseg_long);
            seg_nnz = sc.get<uint32_t>(nvirt + 1);
            lv.heavy_nnz[1] = seg_nnz;
            // over-long segments of gathered rows: their records, from their runs (the paths below read records)
            if (pl.ga.runs) {
                uint64_t *xsjob = sc.get<uint64_t>((uint64_t)nseg_long + 1);
                uint64_t *xsjob_tmp = sc.get<uint64_t>(scan_scratch_entries((uint64_t)nseg_long + 1));
                device_exclusive_scan<SegExpandJobs, uint64_t>(SegExpandJobs{p1.long_rows, vrow_off, pl.vrun_off}, nseg_long, xsjob, xsjob_tmp, s);
                const uint64_t job_bound = nh / kExpandJob + nseg_long;
                expand_segments_kernel<T><<<(unsigned)std::min<uint64_t>(job_bound, (uint64_t)ctx->cus * 8), kExpandThreads, 0, s>>>(
                    p1.long_rows, nseg_long, xsjob, vrow_off, pl.vrun_off, pl.vrun_end, pl.ga.runs, pl.ga.b_colidx, pl.ga.b_vals, qstage);
            }
            // first those whose column range is narrow (hub rows): one dense accumulator per column, no sort at all
            // (debugging aid: OSP_DENSE_SEG=0 leaves them to the two paths below)
            const uint32_t *rest_list = p1.long_rows;
            uint32_t nrest = nseg_long;
            uint32_t *hscan = sc.get<uint32_t>((uint64_t)nseg_long + 1);
            uint64_t *sscan_tmp = sc.get<uint64_t>(scan_scratch_entries(nseg_long));  // NOT hscan_tmp: that one is sized for nlong
            {
                const SegDenseFlag df{p1.long_rows, pl.vcol0, pl.vcol1, getenv("OSP_DENSE_SEG") ? atoi(getenv("OSP_DENSE_SEG")) : 1};
                device_exclusive_scan<SegDenseFlag, uint32_t>(df, nseg_long, hscan, (uint32_t *)sscan_tmp, s);
                const uint32_t ndense = d2h(hscan + nseg_long, s);
                if (ndense) {
                    uint32_t *dense_list = sc.get<uint32_t>(ndense), *others = sc.get<uint32_t>(nseg_long - ndense);
                    seg_split_list_kernel<SegDenseFlag><<<grid_for(nseg_long, 256), 256, 0, s>>>(df, hscan, nseg_long, dense_list, others);
                    if (ctx->dense_atomic[sizeof(T) == 8])
                        dense_segment_kernel<T, true><<<grid_for(ndense, kDenseWaves), kDenseWaves * kWave, 0, s>>>(dense_list, ndense, vrow_off, pl.vcol0, pl.vcol1,
                                                                              qstage, seg_nnz);
                    else
                        dense_segment_kernel<T, false><<<grid_for(ndense, kDenseWaves), kDenseWaves * kWave, 0, s>>>(dense_list, ndense, vrow_off, pl.vcol0, pl.vcol1,
                                                                              qstage, seg_nnz);
                    res->info.dense_segments += ndense;
                    rest_list = others;
                    nrest = nseg_long - ndense;
                }
            }
            dbg_sync(s, "over-long segments: dense accumulation");
            // the rest by length: up to kBigTileCap -> one big LDS tile each, reduced in place; beyond -> global sort
            // (debugging aid: OSP_BIGTILE_CAP=0 sends every such segment down the global-sort path)
            const uint32_t big_cap = getenv("OSP_BIGTILE_CAP") ? std::min<uint32_t>((uint32_t)strtoul(getenv("OSP_BIGTILE_CAP"), nullptr, 10), kBigTileCap)
                                                                : (uint32_t)kBigTileCap;
            uint32_t nhuge = 0, nmid = 0;
            uint32_t *huge_list = nullptr, *mid_list = nullptr;
            if (nrest) {
                const SegHugeFlag hf{rest_list, vrow_off, big_cap};
                device_exclusive_scan<SegHugeFlag, uint32_t>(hf, nrest, hscan, (uint32_t *)sscan_tmp, s);
                nhuge = d2h(hscan + nrest, s);
                nmid = nrest - nhuge;
                huge_list = sc.get<uint32_t>(nhuge);
                mid_list = sc.get<uint32_t>(nmid);
                seg_partition_kernel<<<grid_for(nrest, 256), 256, 0, s>>>(hf, hscan, nrest, huge_list, mid_list);
            }
            res->info.sorted_segments += nhuge;
            if (nmid) {
                TileDesc *bdesc = sc.get<TileDesc>(nmid);
                seg_tile_desc_kernel<<<grid_for(nmid, 256), 256, 0, s>>>(mid_list, nmid, vrow_off, pl.vcol0, pl.vcol1, bdesc);
                uint32_t *bticket = sc.get<uint32_t>(1);
                zero_async(s, {{bticket, sizeof(uint32_t)}});
                const uint32_t bgrid = ctx->cus * (uint32_t)merge_wgs_per_cu<T, kBigTileThreads, kBigTileCap>();
                OSP_WITH_RA(ctx, merge_tiles_kernel<T, kBigTileThreads, 32, kBigTileCap, kMergeMaxWgs, RA>
                            <<<std::min<uint32_t>(nmid, bgrid), kBigTileThreads, 0, s>>>(bdesc, nmid, lv, colbits, nullptr, bticket, nullptr, nullptr,
                                                                                         nullptr, nullptr));
            }
            dbg_sync(s, "over-long segments: big in-place tiles");
            if (nhuge) {
                // one output entry fed by more products than any tile holds: global stable sort on (segment, col),
                // run sums in place in the second buffer
                uint64_t *soff = sc.get<uint64_t>((uint64_t)nhuge + 1);
                device_exclusive_scan<HeavyLen, uint64_t>(HeavyLen{huge_list, vrow_off}, nhuge, soff, sscan_tmp, s);
                const uint64_t ns = d2h(soff + nhuge, s);
                res->info.sorted_partials += ns;
                uint64_t *keys[2] = {sc.get<uint64_t>(ns + 1), sc.get<uint64_t>(ns + 1)};  // +1: the idle one holds the run heads later
                uint32_t *poss[2] = {sc.get<uint32_t>(ns), sc.get<uint32_t>(ns)};
                uint32_t *hist = sc.get<uint32_t>(sort_hist_entries(ns));
                uint32_t *hist_tmp = sc.get<uint32_t>(scan_scratch_entries(sort_hist_entries(ns)));
                heavy_fill_kernel<<<grid_for(ns, 256), 256, 0, s>>>(huge_list, soff, nhuge, vrow_off, 0, colbits, (const char *)qstage,
                                                                    (uint32_t)sizeof(Part<T>), ns, keys[0], poss[0]);
                const int cur = device_radix_sort_pairs<uint64_t>(keys, poss, ns, colbits + bits_for(nhuge), hist, hist_tmp, s, ctx->rank_atomic);
                T *sorted_val = sc.get<T>(ns);
                heavy_gather_kernel<T><<<grid_for(ns, 256), 256, 0, s>>>(poss[cur], qstage, ns, sorted_val);
                uint64_t *headscan = sc.get<uint64_t>(ns + 1);
                uint64_t *headscan_tmp = sc.get<uint64_t>(scan_scratch_entries(ns));
                device_exclusive_scan<HeavyHeadFlag, uint64_t>(HeavyHeadFlag{keys[cur]}, ns, headscan, headscan_tmp, s);
                uint64_t *head_pos = keys[cur ^ 1];  // the idle key buffer: one entry per run and a sentinel, <= ns + 1
                heavy_heads_kernel<<<grid_for(ns + 1, 256), 256, 0, s>>>(keys[cur], headscan, ns, head_pos);
                heavy_reduce_kernel<T><<<grid_for(ns, 256), 256, 0, s>>>(keys[cur], sorted_val, headscan, head_pos, ns, huge_list, soff,
                                                                         nhuge, vrow_off, 0, colbits, qstage);
                heavy_rows_kernel<<<grid_for(nhuge, 256), 256, 0, s>>>(huge_list, soff, nhuge, headscan, seg_nnz);
            }
            dbg_sync(s, "over-long segments: global sort");
            heavy_src_inplace_kernel<<<grid_for(nseg_long, 256), 256, 0, s>>>(p1.long_rows, nseg_long, vrow_off, 0, seg_src);
        }
        lv.heavy_nnz[1] = seg_nnz;
        // ---- one chain: every long row's placeholder is replaced by the tiles of its segments ----
        uint32_t *j0 = sc.get<uint32_t>(nlong), *tb = sc.get<uint32_t>((uint64_t)nlong + 1), *extra = sc.get<uint32_t>((uint64_t)nlong + 1);
        chain_rows_kernel<<<grid_for(nlong + 1, 256), 256, 0, s>>>(p0.long_rows, nlong, p0.tile_rows, p0.ntiles, vbase, p1.tile_rows,
                                                                  p1.ntiles, j0, tb, extra);
        device_exclusive_scan<LoadU32, uint32_t>(LoadU32{extra}, nlong, extra, (uint32_t *)hscan_tmp, s);
        ntot = p0.ntiles + (p1.ntiles - nlong);   // (the scan's total: every long row's tiles but one -- no read-back)
        desc = sc.get<TileDesc>(ntot);
        tile_desc_kernel<(int)kCap><<<grid_for(p0.ntiles, 256), 256, 0, s>>>(p0.tile_rows, p0.ntiles, r1, io.row_off, base, 0u, j0, extra,
                                                                           nlong, tb, pl.vcol0, pl.vcol1, desc, nullptr, nullptr, io.rowfirst0);
        tile_desc_kernel<(int)kCap><<<grid_for(p1.ntiles, 256), 256, 0, s>>>(p1.tile_rows, p1.ntiles, nvirt, vrow_off, 0, 1u, j0, extra,
                                                                           nlong, tb, pl.vcol0, pl.vcol1, desc, pl.ga.runs ? pl.vrun_off : nullptr,
                                                                           pl.vrun_end);
        dbg_sync(s, "tile chain");
    } else {
        desc = sc.get<TileDesc>(ntot);
        tile_desc_kernel<(int)kCap><<<grid_for(p0.ntiles, 256), 256, 0, s>>>(p0.tile_rows, p0.ntiles, r1, io.row_off, base, 0u, nullptr,
                                                                           nullptr, 0u, nullptr, nullptr, nullptr, desc, nullptr, nullptr, io.rowfirst0);
    }
    uint64_t *tile_status = sc.get<uint64_t>(ntot);
    // ticket counters: one word (the kernel also takes several plus an arrival counter -- osp_kernels.h, take_ticket; measured in
    // round 3: no gain while the look-back is on -- the chain and the hash count bound the kernel, not the word; the switch is gone)
    const uint32_t nshards = 1u;
    uint32_t *ticket = sc.get<uint32_t>((uint64_t)(nshards + 1) * kTicketStride);
    zero_async(s, {{tile_status, (uint64_t)ntot * sizeof(uint64_t)}, {ticket, (uint64_t)(nshards + 1) * kTicketStride * sizeof(uint32_t)}});
    dbg_sync(s, "tile planning, splits, over-long segments");
    tm.begin(PH_MERGE_K);
    // persistent workgroups: as many as the LDS lets run at once
    if (io.ct.enabled) {
        const uint32_t rw_grid = ctx->cus * (uint32_t)merge_wgs_per_cu<T, kMergeThreads, TileCap<T>::value, kMergeMaxWgs, 64>();
        OSP_WITH_RA(ctx, merge_tiles_kernel<T, kMergeThreads, 64, TileCap<T>::value, kMergeMaxWgs, RA>
                    <<<std::min<uint32_t>(ntot, rw_grid), kMergeThreads, 0, s>>>(desc, ntot, lv, colbits, tile_status, ticket, io.out_in, io.c_col,
                                                                                 io.c_val, io.out_out, io.ct, nshards, io.abort_word));
    } else if (pl.ga.runs || io.runs0) {   // the panel has gathered rows: the instantiation that forms their records
        GatherArgs<T> ga = pl.ga;
        ga.runs0 = io.runs0;
        if (io.runs0) { ga.b_colidx = io.b_colidx; ga.b_vals = io.b_vals; }
        const uint32_t merge_grid = ctx->cus * (uint32_t)merge_wgs_per_cu<T, kMergeThreads, TileCap<T>::value>();
        OSP_WITH_RA(ctx, merge_tiles_kernel<T, kMergeThreads, 0, TileCap<T>::value, kMergeMaxWgs, RA, true>
                    <<<std::min<uint32_t>(ntot, merge_grid), kMergeThreads, 0, s>>>(desc, ntot, lv, colbits, tile_status, ticket, io.out_in, io.c_col,
                                                                                    io.c_val, io.out_out, ChunkTable<T>{}, nshards, io.abort_word, ga));
    } else {
        const uint32_t merge_grid = ctx->cus * (uint32_t)merge_wgs_per_cu<T, kMergeThreads, TileCap<T>::value>();
        OSP_WITH_RA(ctx, merge_tiles_kernel<T, kMergeThreads, 0, TileCap<T>::value, kMergeMaxWgs, RA>
                    <<<std::min<uint32_t>(ntot, merge_grid), kMergeThreads, 0, s>>>(desc, ntot, lv, colbits, tile_status, ticket, io.out_in, io.c_col,
                                                                                    io.c_val, io.out_out, ChunkTable<T>{}, nshards, io.abort_word));
    }
    tm.end(PH_MERGE_K);
    dbg_sync(s, "merge tiles");
#ifdef OSP_MERGE_PROF
    if (getenv("OSP_VERBOSE")) {   // the tiles by sort passes (key bits) and fill
        std::vector<TileDesc> hd(ntot);
        copy_d2h(hd.data(), desc, (size_t)ntot * sizeof(TileDesc), s);
        uint64_t byp[5] = {0}, ent[5] = {0}, gt = 0, ge = 0, gr = 0, bits[33] = {0};
        for (const TileDesc &t : hd) {
            if (t.n > kCap) continue;
            int rb = 0; while ((1u << rb) < t.nr) rb++;
            const int kb = t.kbits ? (int)t.kbits : colbits + rb;
            const int np = std::min(4, (kb + kDigitBits - 1) / kDigitBits);
            byp[np]++; ent[np] += t.n; bits[std::min(kb, 32)] += 1;
            if (t.rcnt) { gt++; ge += t.n; gr += t.rcnt; }
        }
        fprintf(stderr, "[osp]   tiles by sort passes:");
        for (int k = 0; k < 5; k++) if (byp[k]) fprintf(stderr, " %d passes: %llu tiles, %.0f entries each;", k, (unsigned long long)byp[k], (double)ent[k] / byp[k]);
        fprintf(stderr, " gathered: %llu tiles, %.0f entries and %.1f runs each\n[osp]   tiles by key bits:", (unsigned long long)gt, gt ? (double)ge / gt : 0.0, gt ? (double)gr / gt : 0.0);
        for (int k = 0; k <= 32; k++) if (bits[k]) fprintf(stderr, " %d: %llu;", k, (unsigned long long)bits[k]);
        fprintf(stderr, "\n");
    }
    if (getenv("OSP_VERBOSE")) {   // (`make prof`: cycles of thread 0 of every workgroup between the kernel's marks)
        unsigned long long hp[16] = {0}, z[16] = {0};
        OSP_HIP(hipStreamSynchronize(s));
        (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(osp_merge_prof), sizeof(hp));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(osp_merge_prof), z, sizeof(z));
        static const char *names[12] = {"run table / hash init", "keys + hash count", "publish", "rank", "digit scan", "scatter", "values + heads + scan",
                                        "look-back + ticket", "run sums", "compaction", "output + next tile's request", "gathered: lookup + columns"};
        double tot = 0;
        for (int k = 0; k < 12; k++) tot += (double)hp[k];
        fprintf(stderr, "[osp]   merge kernel, %u tiles, cycles of thread 0 per phase:", ntot);
        for (int k = 0; k < 12; k++) fprintf(stderr, " %s %.1f %%;", names[k], 100.0 * hp[k] / tot);
        fprintf(stderr, " %.0f cycles per tile and workgroup\n", tot / std::max(1u, ntot));
    }
#endif
    res->info.merge_launches++;
    if (p1.nlong) {
        heavy_copy_kernel<T><<<grid_for(p1.nlong, 8), 256, 0, s>>>(p1.long_rows, p1.nlong, seg_src, seg_nnz, vptr, qstage, nullptr, nullptr,
                                                                   io.c_col, io.c_val);
        heavy_copy_rest_kernel<T><<<p1.nlong, 256, 0, s>>>(p1.long_rows, p1.nlong, seg_src, seg_nnz, vptr, qstage, nullptr, nullptr,
                                                           io.c_col, io.c_val);
    }
    chain_finish_kernel<<<grid_for(std::max<uint32_t>(p0.nlong, 1), 256), 256, 0, s>>>(p0.long_rows, p0.nlong, vbase, vptr, io.out_out, r1,
                                                                                       io.c_rowptr);
    dbg_sync(s, "copy of reduced segments, chain finish");
    OSP_HIP(hipGetLastError());  // a rejected launch must not pass silently
}

// Streaming mode (osp_spgemm_csc_csr_panels): every row panel is handed to the caller as soon as it is merged and
// its buffers are reused for the next one -- C is never resident as a whole.
struct PanelSink {
    osp_panel_fn fn;
    void *user;
};

// ---- stages shared by both entry points: partial products of each row -> final CSR ----------------
constexpr uint64_t kPartialsOnDevice = ~0ull;   // merge_pipeline's P: not read back yet, it is d_row_off[M_all]
template <class T>
static void merge_pipeline(Context *ctx, Result *res, Producer<T> &prod, uint64_t M_all, uint64_t N,
                           const uint64_t *d_row_off,
                           uint64_t P, uint64_t cap_cfg, PhaseTimer &tm, uint64_t r_lo = 0, uint64_t r_hi = ~0ull,
                           uint64_t off_lo = 0, const PanelSink *sink = nullptr, const ChunkTable<T> *ct = nullptr,
                           const DirectSrc *ds = nullptr, const std::vector<uint64_t> *cuts = nullptr, const ShortRuns<T> *sr = nullptr) {
    // cuts (optional, ascending row ids inside (r_lo, r_hi)): a panel never reaches across one of them -- the multi-GPU
    // merge makes its panels end where the pieces it receives end (osp_multi.h)
    // output rows [r_lo, r_hi) only (row-sharded multi-GPU mode); P = their partial products, off_lo = row_off[r_lo]
    if (r_hi == ~0ull) r_hi = M_all;
    const uint64_t M = r_hi - r_lo;
    res->info.M = M;
    res->info.row_begin = r_lo;
    res->info.row_end = r_hi;
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    // (The LDS tiles are merged by a stable LSD radix sort; pairwise merging of the pre-sorted chunks was built and
    // measured slower -- osp_merge_runs.h, tools/bench_merge -- and is not wired into the library.)
    const int colbits = std::max(1, bits_for(N));

    // ---- final CSR arrays at an upper bound: nnz(C) <= sum_i min(U_i, N) -------------------------------
    const uint64_t E = 4 + sizeof(T);
    uint64_t cap_c;
    uint64_t *ub = sc.get<uint64_t>(M + 1);  // exclusive scan of the per-row bounds (kept: streaming sizes panels with it)
    {
        Scratch us(ctx);
        uint64_t *ub_tmp = us.get<uint64_t>(scan_scratch_entries(M + 1));
        if (P != 0) device_exclusive_scan<RowUpperBound, uint64_t>(RowUpperBound{d_row_off + r_lo, N}, M, ub, ub_tmp, s);
        if (P == 0) {
            cap_c = 0;
        } else if (P == kPartialsOnDevice) {   // the caller left the count of partial products on the device: one wait for both
            Gather g(s);
            g.add(&P, d_row_off + M_all);
            g.add(&cap_c, (const uint64_t *)ub + M);
            g.wait();
            res->info.partials = P;
        } else {
            cap_c = d2h(ub + M, s);
        }
    }
    if (!sink) res->rowptr = (int64_t *)ctx->alloc((M + 1) * sizeof(int64_t));
    if (P == 0) {
        res->info.nnz_c = 0;
        if (sink) {
            // one empty panel, so that the caller sees every row exactly once
            int64_t *zr = sc.get<int64_t>(M + 1);
            OSP_HIP(hipMemsetAsync(zr, 0, (M + 1) * sizeof(int64_t), s));
            OSP_HIP(hipStreamSynchronize(s));
            res->info.panels = 1;
            const osp_panel_t pd{r_lo, r_hi, 0, zr, nullptr, nullptr, 0, 1, {0, 0}};
            if (sink->fn(&pd, sink->user)) throw Error(OSP_ERR_ARG, "panel callback returned non-zero");
            return;
        }
        OSP_HIP(hipMemsetAsync(res->rowptr, 0, (M + 1) * sizeof(int64_t), s));
        return;
    }
    uint32_t *c_col = nullptr;
    T *c_val = nullptr;
    if (!sink) {
        res->colidx = (uint32_t *)ctx->alloc(std::max<uint64_t>(cap_c, 1) * sizeof(uint32_t));
        res->vals = ctx->alloc(std::max<uint64_t>(cap_c, 1) * sizeof(T));
        c_col = res->colidx;
        c_val = (T *)res->vals;
    }
    // ---- panels: consecutive rows whose partial products fit the staging capacity --------------
    // What is left after the output is shared by the staging buffer and, for long rows, the split
    // buffer and its temporary output (each up to one panel): budget a third of it, with slack.
    size_t free_b = 0, total_b = 0;
    OSP_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += ctx->pooled_bytes + (ctx->sibling ? ctx->sibling->pooled_bytes : 0);   // (what alloc() can free before it fails)
    uint64_t cap = cap_cfg;
    // streaming: the panel's output buffer (at most one record per partial product) comes out of the same budget
    // what a staged partial product needs: its record in the staging buffer, for 9 of 10 another one in the second buffer,
    // and a few per cent for tile tables and the cells of the direct rows -- 2.0 record sizes; 2.6 budgets 30 % on top of
    // that (3.3 until round 3: R-MAT-22 mild ran as 4 panels, now 3: one panel's planning, launches and read-backs less)
    // (round 5, gathered rows: nothing is written for nine records of ten, but both buffers are still addressed by the rows'
    // positions -- allocated in full -- and the run table is sized by a bound: measured 2.25 record sizes per product; 2.5)
    const double per_record = sink ? 3.7 : (ds && ds->gather) ? 2.5 : 2.6;
    if (cap == 0) cap = std::max<uint64_t>((uint64_t)(free_b * 0.85 / (per_record * E)), 1ull << 20);
    cap = std::min<uint64_t>(cap, 0xfffffff0ull);  // staging positions are u32
    if (getenv("OSP_VERBOSE"))
        fprintf(stderr, "[osp] M=%llu N=%llu P=%llu nnzC<=%llu (%.1f GB) free %.1f GB -> staging capacity %llu partial products (%.1f GB)\n",
                (unsigned long long)M, (unsigned long long)N, (unsigned long long)P, (unsigned long long)cap_c, cap_c * E / 1e9,
                free_b / 1e9, (unsigned long long)cap, cap * E / 1e9);
    std::vector<uint64_t> bounds{r_lo};  // absolute row ids
    std::vector<uint64_t> boff{off_lo};  // row_off at the bounds (multi-panel only)
    if (P <= cap && !(cuts && !cuts->empty())) {
        bounds.push_back(r_hi);
        boff.push_back(off_lo + P);
    } else {
        // The panels' bounds are found on the device by one wave (round 5; until then all M + 1 row offsets came to the host:
        // 33 MB through a fresh vector, 6.5 ms of the headline product with the device idle): at most kMaxPanels of them, the
        // bounds and the offsets there in one small read-back.
        constexpr uint32_t kMaxPanels = 4096;
        const uint32_t ncuts = cuts ? (uint32_t)cuts->size() : 0u;
        uint64_t *d_b = sc.get<uint64_t>(2ull * (kMaxPanels + 1) + 2), *d_cuts = sc.get<uint64_t>(std::max<uint32_t>(ncuts, 1));
        if (ncuts) copy_h2d(d_cuts, cuts->data(), ncuts * sizeof(uint64_t), s);
        panel_bounds_kernel<<<1, kWave, 0, s>>>(d_row_off, r_lo, M, cap, d_cuts, ncuts, kMaxPanels, d_b);
        std::vector<uint64_t> hb(2ull * (kMaxPanels + 1) + 2);
        copy_d2h(hb.data(), d_b, hb.size() * sizeof(uint64_t), s);
        const uint64_t np = hb[0], bad = hb[1];
        if (bad != ~0ull)
            throw Error(OSP_ERR_CAPACITY, "output row " + std::to_string(bad) + " has more partial products than the staging capacity of " +
                        std::to_string(cap) + ", or the product needs more than " + std::to_string(kMaxPanels) + " panels");
        for (uint64_t p = 1; p <= np; p++) { bounds.push_back(hb[2 + p]); boff.push_back(hb[2 + (kMaxPanels + 1) + p]); }
        boff[0] = hb[2 + (kMaxPanels + 1)];
    }
    const bool all_rows = r_lo == 0 && r_hi == M_all;
    const uint32_t npanels = (uint32_t)bounds.size() - 1;
    res->info.panels = npanels;
    // staging offset of panel p's first row, and its number of partial products
    auto panel_base = [&](uint32_t p) { return (npanels == 1) ? off_lo : boff[p]; };
    auto panel_count = [&](uint32_t p) { return (npanels == 1) ? P : boff[p + 1] - boff[p]; };
    uint64_t max_panel = 0, max_rows_panel = 0;
    for (uint32_t p = 0; p < npanels; p++) {
        max_panel = std::max(max_panel, panel_count(p));
        max_rows_panel = std::max(max_rows_panel, bounds[p + 1] - bounds[p]);
    }
    Part<T> *stage = sc.get<Part<T>>(max_panel);
    uint64_t *out_nnz = sc.get<uint64_t>((uint64_t)npanels + 1);  // nnz written before panel p
    uint32_t *abort_word = sc.get<uint32_t>(1);   // raised by a tile whose predecessors never published (merge_tiles_kernel's watchdog)
    zero_async(s, {{out_nnz, sizeof(uint64_t)}, {abort_word, sizeof(uint32_t)}});
    auto check_abort = [&](uint32_t flag) {
        if (flag) throw Error(OSP_ERR_HIP, "the merge made no progress for seconds (a tile's predecessors never published their sizes); "
                                           "with several ticket shards that happens when fewer workgroups than shards ever run side by side");
    };
    // With several panels the plan of panel p+1 (VALU-bound: one workgroup per long row, histograms in LDS) runs on the
    // context's second stream beside the multiply of panel p (bound by its scattered stores, its waves mostly parked): the
    // two share the CUs.  Fork: the second stream waits for everything queued before that multiply (so the buffers the
    // plan takes from the pool are no longer in use by panel p-1's merge); join: the first stream waits for the plan before
    // panel p's merge (whose scratch may be what the plan has just given back).  OSP_PLAN_OVERLAP=0 plans every panel in
    // line, before its own multiply (debugging aid, A/B timing).
    const bool overlap = npanels > 1 && !(getenv("OSP_PLAN_OVERLAP") && atoi(getenv("OSP_PLAN_OVERLAP")) == 0);
    if (overlap) ctx->need_aux();
    struct AuxScope {   // ctx->stream is the second stream while this lives
        Context *c; hipStream_t main;
        explicit AuxScope(Context *ctx_) : c(ctx_), main(ctx_->stream) { c->stream = c->aux; }
        ~AuxScope() {
            if (std::uncaught_exceptions()) (void)hipStreamSynchronize(c->aux);   // the plan's buffers go back to the pool next
            c->stream = main;
        }
    };
    typedef std::unique_ptr<PanelPlan<T>> PlanPtr;
    auto plan_one = [&](uint32_t p, bool beside) -> PlanPtr {
        const uint64_t r0 = bounds[p], r1 = bounds[p + 1];
        const uint64_t base = panel_base(p), count = panel_count(p);
        PlanPtr pl(new PanelPlan<T>(ctx));
        if (beside) {
            AuxScope scope(ctx);
            OSP_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_fork, 0));
            tm.begin(PH_MERGE, ctx->stream);
            plan_panel<T>(ctx, res, tm, *pl, d_row_off, r0, r1, base, count, colbits, ds);
            tm.end(PH_MERGE);
            OSP_HIP(hipEventRecord(ctx->aux_join, ctx->stream));
            res->info.plans_overlapped++;
        } else {
            tm.begin(PH_MERGE);
            plan_panel<T>(ctx, res, tm, *pl, d_row_off, r0, r1, base, count, colbits, ds);
            tm.end(PH_MERGE);
        }
        return pl;
    };
    // the multiply of panel p, then -- beside it -- the plan of panel p+1
    auto multiply_and_plan_next = [&](uint32_t p, PanelPlan<T> &plan, PlanPtr &nxt) {
        const uint64_t r0 = bounds[p], r1 = bounds[p + 1];
        const uint64_t base = panel_base(p), count = panel_count(p);
        if (plan.p0.nlong) plan.qstage = plan.sc.template get<Part<T>>(plan.nh);   // (not before: the plan may be a panel ahead)
        const bool beside = overlap && p + 1 < npanels;
        if (beside) { OSP_HIP(hipEventRecord(ctx->aux_fork, s)); ctx->fork_window = true; ctx->releases_in_fork_window = 0; }
        struct WindowEnd { Context *c; ~WindowEnd() { c->fork_window = false; } } window_end{ctx};
        tm.begin(PH_MUL);
        bool column_major = count != 0 && !(sr && plan.p0.nlong == 0), desc_only = false;   // (short rows gathered, no long row: nothing is staged)
        if (count && plan.xjobbase && plan.xjobs_bound) {
            tm.begin(PH_EXPAND_K);
            expand_rows_kernel<T><<<(unsigned)plan.xjobs_bound, kExpandThreads, 0, s>>>(plan.p0.long_rows, plan.p0.nlong, plan.xjobbase, d_row_off, base, ds->rowfirst,
                                                                                     ds->off, ds->bs, ds->av_in_order ? nullptr : ds->perm, (const T *)ds->a_vals,
                                                                                     ds->b_colidx, (const T *)ds->b_vals, stage);
            tm.end(PH_EXPAND_K);
            res->info.expand_launches++;
            res->info.expand_partials += plan.xpartials;
        }
        if (count && plan.expand_ok) {
            // what is left for the column-major multiply: rows written through cells -- hub rows, direct rows with an over-long range
            desc_only = true;
            // (with every planned row gathered -- the default -- nothing can have been counted: no round trip)
            const uint32_t nw = (plan.nwritten && plan.may_write) ? d2h(plan.nwritten, s) : 0u;
            column_major = plan.hub.cells != nullptr || nw != 0 || (plan.mode_rows[kModeDirect] != 0 && plan.ga.runs == nullptr);   // (a panel whose run table would not fit 32 bits writes its direct rows)
        }
        // (the compacted multiply pays where few chunks are left to write; a panel whose hub rows hold a third of its products
        // walks all of A as before: Graph500 scale 22, 77 % in hub rows, 34.0 against 28-33 ms per launch)
        const bool mostly_hub = plan.hub.cells != nullptr && plan.mode_partials[kModeStretch] * 3 >= count;
        if (column_major && ds) ds->ensure_chunk_off(s);
        if (column_major) prod.produce(r0, r1, npanels == 1 && all_rows, base, count, stage, tm, plan.cells, plan.qstage, plan.hub.cells ? &plan.hub : nullptr,
                                       plan.ga.runs != nullptr && !mostly_hub, plan.p0.nlong != 0, desc_only, mostly_hub);
        tm.end(PH_MUL);
        if (beside) {
            ctx->fork_window = false;
            if (ctx->releases_in_fork_window)
                throw Error(OSP_ERR_HIP, "internal: " + std::to_string(ctx->releases_in_fork_window) + " pooled buffers were released between the fork of the "
                                         "second stream and the next panel's plan (the pool is not stream-aware: see Context::fork_window)");
            nxt = plan_one(p + 1, true);
            OSP_HIP(hipStreamWaitEvent(s, ctx->aux_join, 0));
        }
    };
    if (sink) {
        // ---- streaming: one output buffer sized for the largest panel's bound, reused by every panel ----
        std::vector<uint64_t> h_ub(npanels + 1);
        for (uint32_t p = 0; p <= npanels; p++) h_ub[p] = d2h(ub + (bounds[p] - r_lo), s);
        uint64_t max_out = 1;
        for (uint32_t p = 0; p < npanels; p++) max_out = std::max(max_out, h_ub[p + 1] - h_ub[p]);
        c_col = sc.get<uint32_t>(max_out);
        c_val = sc.get<T>(max_out);
        int64_t *prow = sc.get<int64_t>(max_rows_panel + 1);
        uint64_t *cells = sc.get<uint64_t>(2);  // [0] = 0 (entries before the panel), [1] = entries of the panel
        uint64_t nnz_total = 0;
        PlanPtr cur, nxt;
        for (uint32_t p = 0; p < npanels; p++) {
            const uint64_t r0 = bounds[p], r1 = bounds[p + 1];
            const uint64_t base = panel_base(p);
            if (!cur) cur = plan_one(p, false);
            PanelPlan<T> &plan = *cur;
            multiply_and_plan_next(p, plan, nxt);
            tm.begin(PH_MERGE);
            OSP_HIP(hipMemsetAsync(cells, 0, 2 * sizeof(uint64_t), s));
            MergeIO<T> io{stage, d_row_off, r0, r1, base, prow - r0, c_col, c_val, cells, cells + 1};
            if (ct) io.ct = *ct;
            if (sr) { io.runs0 = sr->runs0; io.rowfirst0 = sr->rowfirst0; io.b_colidx = sr->b_colidx; io.b_vals = sr->b_vals; }
            io.abort_word = abort_word;
            merge_panel<T>(ctx, res, tm, io, colbits, plan);
            tm.end(PH_MERGE);
            uint64_t nnz_p = 0;
            uint32_t aflag = 0;
            { Gather g(s); g.add(&nnz_p, (const uint64_t *)cells + 1); g.add(&aflag, (const uint32_t *)abort_word); g.wait(); }  // synchronises: the panel is complete
            check_abort(aflag);
            nnz_total += nnz_p;
            const osp_panel_t pd{r0, r1, nnz_p, prow, c_col, c_val, p, npanels, {0, 0}};
            ctx->ensure_free(2ull << 30);  // the consumer needs room of its own
            if (sink->fn(&pd, sink->user)) throw Error(OSP_ERR_ARG, "panel callback returned non-zero");
            OSP_HIP(hipStreamSynchronize(s));  // whatever the callback queued on this stream reads the buffers
            cur = std::move(nxt);
        }
        res->info.nnz_c = nnz_total;
        return;
    }

    PlanPtr cur, nxt;
    for (uint32_t p = 0; p < npanels; p++) {
        const uint64_t r0 = bounds[p], r1 = bounds[p + 1];
        const uint64_t base = panel_base(p);
        // ---- plan (unless made beside the previous panel's multiply), multiply (or scatter of CSR parts) ----
        if (!cur) cur = plan_one(p, false);
        PanelPlan<T> &plan = *cur;
        multiply_and_plan_next(p, plan, nxt);
        // ---- merge ----
        tm.begin(PH_MERGE);
        MergeIO<T> io{stage, d_row_off, r0, r1, base, res->rowptr - r_lo, c_col, c_val, out_nnz + p, out_nnz + p + 1};
        if (ct) io.ct = *ct;
        if (sr) { io.runs0 = sr->runs0; io.rowfirst0 = sr->rowfirst0; io.b_colidx = sr->b_colidx; io.b_vals = sr->b_vals; }
        io.abort_word = abort_word;
        merge_panel<T>(ctx, res, tm, io, colbits, plan);
        tm.end(PH_MERGE);
        cur = std::move(nxt);
    }
    uint64_t nnz_total = 0;
    uint32_t aflag = 0;
    { Gather g(s); g.add(&nnz_total, (const uint64_t *)out_nnz + npanels); g.add(&aflag, (const uint32_t *)abort_word); g.wait(); }
    check_abort(aflag);
    res->info.nnz_c = nnz_total;
    // The arrays were sized by the bound sum_i min(U_i, N); a product that compresses leaves their tails unused.  Copying the
    // result into arrays of its exact size gives that memory back -- at the price of reading and writing all of C once more
    // (the cage15 shape: 5 of 48 ms for 11.6 GB of tails).  So the copy is made only where the tails are worth it: more than a
    // tenth of the device's memory (or OSP_COMPACT_MIN_WASTE bytes; 0 = always, as until round 3).  Below that the result
    // keeps its bound-sized arrays until it is destroyed, and they go back to the pool whole.
    const uint64_t waste = (Context::bucket(std::max<uint64_t>(cap_c, 1) * sizeof(T)) - Context::bucket(std::max<uint64_t>(nnz_total, 1) * sizeof(T))) +
                           (Context::bucket(std::max<uint64_t>(cap_c, 1) * sizeof(uint32_t)) - Context::bucket(std::max<uint64_t>(nnz_total, 1) * sizeof(uint32_t)));
    const uint64_t min_waste = getenv("OSP_COMPACT_MIN_WASTE") ? strtoull(getenv("OSP_COMPACT_MIN_WASTE"), nullptr, 10) : (uint64_t)(total_b / 10);
    res->info.output_slack_bytes = waste;
    if (Context::bucket(std::max<uint64_t>(nnz_total, 1) * sizeof(T)) * 10 < Context::bucket(std::max<uint64_t>(cap_c, 1) * sizeof(T)) * 7 &&
        waste >= min_waste) {
        res->info.output_slack_bytes = 0;
        tm.begin(PH_COMPACT);
        uint32_t *nc = (uint32_t *)ctx->alloc(std::max<uint64_t>(nnz_total, 1) * sizeof(uint32_t));
        T *nv = (T *)ctx->alloc(std::max<uint64_t>(nnz_total, 1) * sizeof(T));
        if (nnz_total) {
            OSP_HIP(hipMemcpyAsync(nc, c_col, nnz_total * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            OSP_HIP(hipMemcpyAsync(nv, c_val, nnz_total * sizeof(T), hipMemcpyDeviceToDevice, s));
        }
        ctx->release(res->colidx);
        ctx->release(res->vals);
        res->colidx = nc;
        res->vals = nv;
        tm.end(PH_COMPACT);
    }
}

// ---- outer-product producer ---------------------------------------------------------------------
template <class T> struct OuterProducer : Producer<T> {
    Context *ctx;
    Result *res;
    const int64_t *a_colptr; const uint32_t *a_rowidx; const T *a_vals;
    const int64_t *b_rowptr; const uint32_t *b_colidx; const T *b_vals;
    uint64_t k0, k1; int64_t e0;
    const uint64_t *chunk_off;   // (the planner of direct rows rewrites the entries of its chunks, panel by panel)
    int64_t *a_start; uint32_t *a_cnt; uint64_t *prod; uint64_t *prod_off; uint64_t *scan_tmp;
    bool nothing_staged = false;  // row-wise variant and no row is longer than a tile: the tile kernel does it all
    // panels with gathered rows: the list of A's entries whose chunks are written (null: not prepared -- such a panel walks all of A)
    uint32_t *kscan = nullptr, *kscan_tmp = nullptr, *elist = nullptr, *cscan = nullptr, *klist = nullptr;
    uint64_t nnz = 0;
    bool short_gathered = false;  // the short rows are gathered (their chunks: kChunkSkip): a panel without long rows multiplies nothing
    void produce(uint64_t r0, uint64_t r1, bool whole, uint64_t base, uint64_t count, Part<T> *stage,
                 PhaseTimer &tm, const uint32_t *cells, Part<T> *qstage, const HubArgs *hub = nullptr, bool compact = false, bool has_long = true,
                 bool desc_only = false, bool walk_all = false) override {
        if (nothing_staged) return;
        if (short_gathered && !has_long) return;
        hipStream_t s = ctx->stream;
        const uint64_t nk = k1 - k0;
        const bool ind = (compact || short_gathered) && elist && nnz && !walk_all;
        if (ind) {
            const PanelKeepFlag keep{a_rowidx, chunk_off, e0, (uint32_t)r0, r1, desc_only ? 1u : 0u};
            device_exclusive_scan<PanelKeepFlag, uint32_t>(keep, nnz, kscan, kscan_tmp, s);
            panel_keep_list_kernel<<<grid_for(nnz, 256), 256, 0, s>>>(keep, kscan, nnz, elist);
            const PanelKeepColFlag kcf{a_colptr, e0, k0, kscan};
            device_exclusive_scan<PanelKeepColFlag, uint32_t>(kcf, nk, cscan, kscan_tmp, s);
            panel_keep_columns_kernel<<<grid_for(nk, 256), 256, 0, s>>>(a_colptr, b_rowptr, k0, nk, e0, kscan, cscan, a_start, a_cnt, prod, klist);
        } else {
            panel_columns_kernel<<<grid_for(nk, 256), 256, 0, s>>>(a_colptr, a_rowidx, b_rowptr, k0, nk, (uint32_t)r0, r1,
                                                                   whole ? 1 : 0, a_start, a_cnt, prod);
        }
        device_exclusive_scan<LoadU64, uint64_t>(LoadU64{prod}, nk, prod_off, scan_tmp, s);
        uint64_t nblocks = (count + kMulPerBlock - 1) / kMulPerBlock;
        if (ind) nblocks = std::min<uint64_t>(nblocks, (uint64_t)ctx->cus * 8);   // (strides over the slices: osp_kernels.h)
        dbg_sync(s, "panel columns + scan");
        tm.begin(PH_MUL_K);
        if (ind && hub && hub->cells)
            multiply_kernel<T, 2, true><<<(unsigned)nblocks, kMulThreads, 0, s>>>(a_vals, b_colidx, b_vals, b_rowptr, chunk_off, e0,
                                                                                a_start, a_cnt, prod_off, k0, nk, count, base, stage, cells, qstage, *hub, elist, klist);
        else if (ind && cells)
            multiply_kernel<T, 1, true><<<(unsigned)nblocks, kMulThreads, 0, s>>>(a_vals, b_colidx, b_vals, b_rowptr, chunk_off, e0,
                                                                                a_start, a_cnt, prod_off, k0, nk, count, base, stage, cells, qstage, HubArgs{}, elist, klist);
        else if (ind)
            multiply_kernel<T, 0, true><<<(unsigned)nblocks, kMulThreads, 0, s>>>(a_vals, b_colidx, b_vals, b_rowptr, chunk_off, e0,
                                                                                a_start, a_cnt, prod_off, k0, nk, count, base, stage, nullptr, nullptr, HubArgs{}, elist, klist);
        else if (hub && hub->cells)
            multiply_kernel<T, 2><<<(unsigned)nblocks, kMulThreads, 0, s>>>(a_vals, b_colidx, b_vals, b_rowptr, chunk_off, e0,
                                                                          a_start, a_cnt, prod_off, k0, nk, count, base, stage, cells, qstage, *hub);
        else if (cells)
            multiply_kernel<T, 1><<<(unsigned)nblocks, kMulThreads, 0, s>>>(a_vals, b_colidx, b_vals, b_rowptr, chunk_off, e0,
                                                                          a_start, a_cnt, prod_off, k0, nk, count, base, stage, cells, qstage);
        else
            multiply_kernel<T, 0><<<(unsigned)nblocks, kMulThreads, 0, s>>>(a_vals, b_colidx, b_vals, b_rowptr, chunk_off, e0,
                                                                          a_start, a_cnt, prod_off, k0, nk, count, base, stage, nullptr, nullptr);
        tm.end(PH_MUL_K);
        dbg_sync(s, "multiply");
        res->info.multiply_launches++;
    }
};

template <class T> struct PartsProducer : Producer<T> {
    Context *ctx;
    const int64_t *const *d_rowptrs; const uint32_t *const *d_colidxs; const T *const *d_valss;
    int nparts;
    const uint64_t *row_off;
    void produce(uint64_t r0, uint64_t r1, bool, uint64_t base, uint64_t, Part<T> *stage, PhaseTimer &, const uint32_t *, Part<T> *, const HubArgs *, bool, bool, bool, bool) override {
        const uint64_t nr = r1 - r0;
        parts_scatter_kernel<T><<<grid_for(nr * kWave, 256), 256, 0, ctx->stream>>>(d_rowptrs, d_colidxs, d_valss, nparts,
                                                                                    r0, r1, row_off, base, stage);
    }
};

template <class T> struct RecordPartsProducer : Producer<T> {
    Context *ctx;
    const int64_t *const *d_rowptrs; const Part<T> *const *d_recs;
    int nparts;
    const uint64_t *row_off;
    const std::function<void(uint64_t, uint64_t)> *before = nullptr;   // called with the panel's rows before its records are read
    void produce(uint64_t r0, uint64_t r1, bool, uint64_t base, uint64_t, Part<T> *stage, PhaseTimer &, const uint32_t *, Part<T> *, const HubArgs *, bool, bool, bool, bool) override {
        if (before) (*before)(r0, r1);
        const uint64_t nr = r1 - r0;
        parts_scatter_rec_kernel<T><<<grid_for(nr * kWave, 256), 256, 0, ctx->stream>>>(d_rowptrs, d_recs, nparts, r0, r1, row_off, base, stage);
    }
};

