// osp_multi.h -- the k-sharded product over several GPUs of one node, inside the library (SURVEY.md 8e, 8b "optional
// device ordinal list").  Included by osp_api.hip (same translation unit: it uses the context, the scratch pool, the
// producers and the merge pipeline defined there).
//
// The reference is one process and one thread (SURVEY.md sections 2, 5); there is nothing to mirror.  Shape:
//   * k is cut into G slabs of equal partial-product count; rank g holds ONLY columns [k_g, k_g+1) of A and rows
//     [k_g, k_g+1) of B (osp_multi_operands_create uploads them once).
//   * every rank runs the symbolic phase of its slab -> records per output row; the per-row counts of all ranks are
//     summed on rank 0 and the rows are cut into G ranges of equal exchanged volume, each range into R sub-panels.
//   * PIPELINE.  Rank g multiplies sub-panel by sub-panel: for r = 0.., the piece of sub-panel r it owes every OTHER rank
//     (destination (g+1) mod G first, so that at any time the ranks start on different destinations) into that
//     destination's send slot, then its own piece straight into its receive buffer.  Every destination has a copy stream
//     and two send slots of its own: the piece leaves (hipMemcpyPeerAsync: the xGMI DMA engines) as soon as it is multiplied,
//     on the link to ITS owner, while the pieces for the other destinations and the next sub-panel multiply -- up to G-1
//     copies of one rank are in flight at once, one per link.  The owner merges sub-panel r on a stream (and pool) of its
//     own as soon as its G pieces have arrived, while the later sub-panels are still multiplied and copied.
//     Time at G ranks ~ max(multiply + merge of P/G, exchange of (G-1)/G * P/G per rank over G-1 links) instead of their sum.
//   * the merge sums the G pieces of a row in rank order = ascending k, the single-GPU order: bit-identical results.
// One host thread per rank drives its GPU (the merge's planning reads scalars back and would otherwise serialise the
// ranks); they meet at a few host barriers and hand each other HIP events.
#pragma once
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace osp {

struct HostBarrier {
    std::mutex m;
    std::condition_variable cv;
    int n, count = 0;
    uint64_t gen = 0;
    explicit HostBarrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> l(m);
        const uint64_t g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); }
        else cv.wait(l, [&] { return gen != g; });
    }
};

struct MultiContext {
    std::vector<int> devices;          // HIP ordinal of every rank (an ordinal may repeat: logical ranks sharing a GPU)
    std::vector<Context *> ctx;        // one context (stream + pool) per rank: symbolic phase, multiply
    std::vector<Context *> mctx;       // a second one per rank for the merge of what arrives: it runs beside the multiply
    std::vector<std::vector<hipStream_t>> copy;   // copy[g][h]: the stream of rank g's copies to rank h (one per link)
    ~MultiContext() {
        for (size_t g = 0; g < ctx.size(); g++) {
            if (!ctx[g]) continue;
            (void)hipSetDevice(ctx[g]->device);
            if (g < copy.size())
                for (hipStream_t cs : copy[g]) if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
            if (g < mctx.size() && mctx[g]) osp_context_destroy((osp_context_t)mctx[g]);
            osp_context_destroy((osp_context_t)ctx[g]);
        }
    }
};

// the slabs, resident on their ranks
struct MultiOperands {
    MultiContext *mc = nullptr;
    int dtype = OSP_F64;
    uint64_t M = 0, K = 0, N = 0, partials = 0;
    std::vector<uint64_t> k_bounds;                  // G + 1
    struct Slab { int64_t *a_colptr = nullptr; uint32_t *a_rowidx = nullptr; void *a_vals = nullptr;
                  int64_t *b_rowptr = nullptr; uint32_t *b_colidx = nullptr; void *b_vals = nullptr;
                  uint64_t K = 0, nnz_a = 0, nnz_b = 0; };
    std::vector<Slab> slab;
    float ms_upload = 0;
    ~MultiOperands() {
        if (!mc) return;
        for (size_t g = 0; g < slab.size(); g++) {
            Context *c = mc->ctx[g];
            c->release(slab[g].a_colptr); c->release(slab[g].a_rowidx); c->release(slab[g].a_vals);
            c->release(slab[g].b_rowptr); c->release(slab[g].b_colidx); c->release(slab[g].b_vals);
        }
    }
};

struct MultiResult {
    MultiContext *mc = nullptr;
    int dtype = OSP_F64;
    std::vector<Result *> shard;          // rank g: rows [row_bounds[g], row_bounds[g + 1])
    std::vector<uint64_t> row_bounds;
    osp_multi_info_t info{};
    ~MultiResult() { for (Result *r : shard) destroy_result(r); }
};

// out[r] = sum over the G arrays (all on this device) of in_g[r]
__global__ void sum_offsets_kernel(const uint64_t *const *in, int G, uint64_t n, uint64_t *out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint64_t s = 0;
    for (int g = 0; g < G; g++) s += in[g][r];
    out[r] = s;
}
__global__ void gather_u64_kernel(const uint64_t *src, const uint64_t *idx, uint32_t n, uint64_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
// rp[i] -= first   (a slice of a rank's record offsets becomes the offsets of one received piece)
__global__ void rebase_offsets_kernel(int64_t *rp, uint64_t n, int64_t first) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rp[i] -= first;
}

// symbolic phase of one slab (the rank's columns of A / rows of B as arrays of their own): records per output row and
// the offset of every chunk.  The same launches spgemm_impl makes, without the variants a slab does not use.
template <class T>
static uint64_t slab_symbolic(Context *ctx, uint64_t M, uint64_t Ks, const int64_t *a_colptr, const uint32_t *a_rowidx, const int64_t *b_rowptr,
                              uint64_t nnz, uint64_t *row_off, uint64_t *chunk_off) {
    hipStream_t s = ctx->stream;
    if (nnz == 0) {
        OSP_HIP(hipMemsetAsync(row_off, 0, (M + 1) * sizeof(uint64_t), s));
        return 0;
    }
    Scratch ss(ctx);
    uint32_t *ka = ss.get<uint32_t>(nnz), *pa = ss.get<uint32_t>(nnz), *kb = ss.get<uint32_t>(nnz), *pb = ss.get<uint32_t>(nnz);
    uint32_t *rows_sorted = ss.get<uint32_t>(nnz), *perm = ss.get<uint32_t>(nnz), *w_sorted = ss.get<uint32_t>(nnz);
    uint32_t *rowfirst = ss.get<uint32_t>(M + 1);
    uint32_t *hist = ss.get<uint32_t>(rs_hist_entries(nnz));
    uint32_t *hist_tmp = ss.get<uint32_t>(scan_scratch_entries(rs_hist_entries(nnz)));
    uint64_t *offs_sorted = ss.get<uint64_t>(nnz + 1);
    uint64_t *scan_tmp = ss.get<uint64_t>(scan_scratch_entries(std::max<uint64_t>(nnz, M + 1)));
    uint32_t *w = ss.get<uint32_t>(nnz);
    sym_chunk_len_kernel<<<grid_for(nnz, 256), 256, 0, s>>>(a_colptr, b_rowptr, 0, Ks, 0, nnz, w, nullptr);
    device_sort_rows<SymEpilogue>(a_rowidx, nnz, std::max(1, bits_for(M)), ka, pa, kb, pb, hist, hist_tmp,
                                  SymEpilogue{w, nullptr, rows_sorted, perm, w_sorted, nullptr}, s, ctx->rank_atomic);
    device_exclusive_scan<LoadU32As64, uint64_t>(LoadU32As64{w_sorted}, nnz, offs_sorted, scan_tmp, s);
    sym_row_offsets_kernel<<<grid_for(M + 1, 256), 256, 0, s>>>(rows_sorted, offs_sorted, nnz, M, row_off, rowfirst);
    sym_scatter_offsets_kernel<<<grid_for(nnz, 256), 256, 0, s>>>(perm, offs_sorted, rows_sorted, row_off, 0, nnz, chunk_off);
    return d2h(offs_sorted + nnz, s);
}

// everything the rank threads of one product share
template <class T> struct MultiShared {
    int G = 0, R = 1;
    HostBarrier bar;
    std::atomic<int> failed{0};
    std::mutex err_m;
    int err_status = OSP_OK;
    std::string err_msg;
    std::vector<uint64_t *> row_off;              // per rank: device array on that rank, M + 1 record offsets
    std::vector<uint64_t> P;                      // per rank
    std::vector<uint64_t> bounds;                 // G * R + 1 row bounds
    std::vector<std::vector<uint64_t>> offs;      // offs[g][j] = row_off_g[bounds[j]]
    std::vector<std::vector<Part<T> *>> recv;     // recv[h][g]: records of range h computed by rank g (on rank h)
    std::vector<std::vector<int64_t *>> rp;       // rp[h][g]: their per-row offsets (M_h + 1)
    // arrival of sub-panel r of range h from rank g: a HIP event on the sender's stream, announced through `arrived`
    std::vector<hipEvent_t> ev;                   // [(h * G + g) * R + r]
    std::vector<int> arrived;                     // [h * R + r]: how many of the G events have been recorded
    std::mutex arr_m;
    std::condition_variable arr_cv;
    explicit MultiShared(int g, int r) : G(g), R(r), bar(g) {}
    void fail(int status, const std::string &msg) {
        std::lock_guard<std::mutex> l(err_m);
        if (!failed.exchange(1)) { err_status = status; err_msg = msg; }
        arr_cv.notify_all();
    }
};

template <class T>
static void multi_rank_main(MultiContext *mc, const MultiOperands *ops, MultiShared<T> *sh, MultiResult *out, const osp_config_t cfg, int g) {
    Context *ctx = mc->ctx[g];
    const int G = sh->G, R = sh->R;
    const uint64_t M = ops->M, N = ops->N;
    // a failing rank keeps walking to every barrier (doing nothing), so that nobody waits for it for ever
    auto guarded = [&](auto &&fn) {
        if (sh->failed.load()) return;
        try { fn(); }
        catch (const Error &e) { sh->fail(e.status, "rank " + std::to_string(g) + ": " + e.what()); }
        catch (const std::exception &e) { sh->fail(OSP_ERR_ALLOC, "rank " + std::to_string(g) + ": " + e.what()); }
    };
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    Scratch sc(ctx);
    PhaseTimer tm(s);
    const MultiOperands::Slab &sl = ops->slab[g];
    uint64_t *row_off = nullptr, *chunk_off = nullptr;
    OuterProducer<T> prod;
    Result dummy;   // the producer counts its launches into a result
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
        return (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    // ---- A. symbolic phase of the slab ----
    guarded([&] {
        row_off = sc.get<uint64_t>(M + 1);
        chunk_off = sc.get<uint64_t>(sl.nnz_a);
        sh->P[g] = slab_symbolic<T>(ctx, M, sl.K, sl.a_colptr, sl.a_rowidx, sl.b_rowptr, sl.nnz_a, row_off, chunk_off);
        sh->row_off[g] = row_off;
        prod.ctx = ctx; prod.res = &dummy;
        prod.a_colptr = sl.a_colptr; prod.a_rowidx = sl.a_rowidx; prod.a_vals = (const T *)sl.a_vals;
        prod.b_rowptr = sl.b_rowptr; prod.b_colidx = sl.b_colidx; prod.b_vals = (const T *)sl.b_vals;
        prod.k0 = 0; prod.k1 = sl.K; prod.e0 = 0; prod.chunk_off = chunk_off;
        prod.a_start = sc.get<int64_t>(sl.K); prod.a_cnt = sc.get<uint32_t>(sl.K);
        prod.prod = sc.get<uint64_t>(sl.K); prod.prod_off = sc.get<uint64_t>(sl.K + 1);
        prod.scan_tmp = sc.get<uint64_t>(scan_scratch_entries(sl.K));
        OSP_HIP(hipStreamSynchronize(s));
    });
    const float ms_symbolic = ms_since(t_begin);
    sh->bar.wait();
    // ---- B. rank 0: total records per row over all ranks -> row ranges of equal exchanged volume, R sub-panels each ----
    if (g == 0) guarded([&] {
        Scratch cs0(ctx);
        std::vector<const uint64_t *> h_in(G);
        for (int q = 0; q < G; q++) {
            uint64_t *tmp = cs0.get<uint64_t>(M + 1);
            OSP_HIP(hipMemcpyPeerAsync(tmp, ctx->device, sh->row_off[q], mc->ctx[q]->device, (M + 1) * sizeof(uint64_t), s));
            h_in[q] = tmp;
        }
        const uint64_t **d_in = (const uint64_t **)cs0.get<void *>(G);
        copy_h2d(d_in, h_in.data(), G * sizeof(void *), s);
        uint64_t *tot = cs0.get<uint64_t>(M + 1);
        sum_offsets_kernel<<<grid_for(M + 1, 256), 256, 0, s>>>(d_in, G, M + 1, tot);
        const uint32_t nb = (uint32_t)(G * R);
        uint64_t *d_b = cs0.get<uint64_t>(2ull * (nb + 1));
        shard_bounds_kernel<<<grid_for(nb + 1, 64), 64, 0, s>>>(tot, tot, M, nb, d_b, d_b + nb + 1);
        sh->bounds.resize(nb + 1);
        copy_d2h(sh->bounds.data(), d_b, (nb + 1) * sizeof(uint64_t), s);
        sh->bounds[0] = 0; sh->bounds[nb] = M;
        for (uint32_t j = 1; j <= nb; j++) sh->bounds[j] = std::max(sh->bounds[j], sh->bounds[j - 1]);
    });
    sh->bar.wait();
    // ---- C. this rank's record offsets at the bounds ----
    const uint32_t nb = (uint32_t)(G * R);
    guarded([&] {
        Scratch cs1(ctx);
        uint64_t *d_idx = cs1.get<uint64_t>(nb + 1), *d_o = cs1.get<uint64_t>(nb + 1);
        copy_h2d(d_idx, sh->bounds.data(), (nb + 1) * sizeof(uint64_t), s);
        gather_u64_kernel<<<grid_for(nb + 1, 64), 64, 0, s>>>(row_off, d_idx, nb + 1, d_o);
        sh->offs[g].resize(nb + 1);
        copy_d2h(sh->offs[g].data(), d_o, (nb + 1) * sizeof(uint64_t), s);
    });
    sh->bar.wait();
    // ---- D. receive buffers of this rank's row range, and the offsets of the pieces it will receive ----
    const uint64_t rb0 = sh->failed.load() ? 0 : sh->bounds[(size_t)g * R], rb1 = sh->failed.load() ? 0 : sh->bounds[(size_t)(g + 1) * R];
    const uint64_t Mh = rb1 - rb0;
    std::vector<void *> pooled;   // released when the product is done
    uint64_t recv_total = 0;
    guarded([&] {
        for (int q = 0; q < G; q++) {
            const uint64_t cnt = sh->offs[q][(size_t)(g + 1) * R] - sh->offs[q][(size_t)g * R];
            recv_total += cnt;
            sh->recv[g][q] = (Part<T> *)ctx->alloc(std::max<uint64_t>(cnt, 1) * sizeof(Part<T>));
            pooled.push_back(sh->recv[g][q]);
            sh->rp[g][q] = (int64_t *)ctx->alloc((Mh + 1) * sizeof(int64_t));
            pooled.push_back(sh->rp[g][q]);
            OSP_HIP(hipMemcpyPeerAsync(sh->rp[g][q], ctx->device, sh->row_off[q] + rb0, mc->ctx[q]->device, (Mh + 1) * sizeof(uint64_t), s));
            rebase_offsets_kernel<<<grid_for(Mh + 1, 256), 256, 0, s>>>(sh->rp[g][q], Mh + 1, (int64_t)sh->offs[q][(size_t)g * R]);
        }
        // the events THIS rank records (an event belongs to the device of the stream that records it; the owner of the rows
        // only waits for it): sub-panel r of the piece it owes rank h
        for (int h = 0; h < G; h++)
            for (int r = 0; r < R; r++) OSP_HIP(hipEventCreateWithFlags(&sh->ev[((size_t)h * G + g) * R + r], hipEventDisableTiming));
        OSP_HIP(hipStreamSynchronize(s));
    });
    sh->bar.wait();
    // ---- E. multiply sub-panel by sub-panel; every finished piece leaves on the stream of ITS destination ----
    uint64_t bytes_sent = 0;
    std::vector<uint64_t> bytes_to(G, 0);
    struct Slot { Part<T> *buf = nullptr; hipEvent_t produced = nullptr, copied = nullptr; bool used = false; };
    std::vector<std::array<Slot, 2>> slots(G);
    struct CopyRec { hipEvent_t t0 = nullptr, t1 = nullptr; };
    std::vector<CopyRec> copies;            // start / end of every peer copy (timestamps: how many overlapped)
    hipEvent_t t_base = nullptr;
    int max_outstanding = 0;
    Context *mctx = mc->mctx[g];
    guarded([&] {
        OSP_HIP(hipEventCreate(&t_base));
        OSP_HIP(hipEventRecord(t_base, s));
        for (int h = 0; h < G; h++) {
            if (h == g) continue;
            uint64_t max_send = 1;
            for (int r = 0; r < R; r++) max_send = std::max(max_send, sh->offs[g][(size_t)h * R + r + 1] - sh->offs[g][(size_t)h * R + r]);
            for (int i = 0; i < std::min(R, 2); i++) {
                slots[h][i].buf = sc.get<Part<T>>(max_send);
                OSP_HIP(hipEventCreateWithFlags(&slots[h][i].produced, hipEventDisableTiming));
                OSP_HIP(hipEventCreateWithFlags(&slots[h][i].copied, hipEventDisableTiming));
            }
        }
        copies.reserve((size_t)G * R);
        for (int r = 0; r < R; r++) {
            // the pieces for the other ranks first (their copies start while the rest of the sub-panel multiplies), rank g's
            // own piece last
            for (int st = 1; st <= G; st++) {
                const int h = (g + st) % G;
                const size_t j = (size_t)h * R + r;
                const uint64_t r0 = sh->bounds[j], r1 = sh->bounds[j + 1];
                const uint64_t base = sh->offs[g][j], count = sh->offs[g][j + 1] - base;
                const uint64_t dst_off = base - sh->offs[g][(size_t)h * R];   // position inside the piece this rank owes rank h
                hipEvent_t arrive = sh->ev[((size_t)h * G + g) * R + r];
                if (h == g) {
                    // own rows: multiplied straight into the receive buffer
                    if (count) prod.produce(r0, r1, false, base, count, sh->recv[g][g] + dst_off, tm, nullptr, nullptr);
                    OSP_HIP(hipEventRecord(arrive, s));
                } else {
                    hipStream_t cs = mc->copy[g][h];
                    if (count) {
                        Slot &sl2 = slots[h][r & 1];
                        if (sl2.used) OSP_HIP(hipStreamWaitEvent(s, sl2.copied, 0));   // the copy out of this slot (sub-panel r - 2) has finished
                        prod.produce(r0, r1, false, base, count, sl2.buf, tm, nullptr, nullptr);
                        OSP_HIP(hipEventRecord(sl2.produced, s));
                        OSP_HIP(hipStreamWaitEvent(cs, sl2.produced, 0));
                        CopyRec cr;
                        OSP_HIP(hipEventCreate(&cr.t0));
                        OSP_HIP(hipEventCreate(&cr.t1));
                        copies.push_back(cr);
                        OSP_HIP(hipEventRecord(cr.t0, cs));
                        OSP_HIP(hipMemcpyPeerAsync(sh->recv[h][g] + dst_off, mc->ctx[h]->device, sl2.buf, ctx->device, count * sizeof(Part<T>), cs));
                        OSP_HIP(hipEventRecord(cr.t1, cs));
                        OSP_HIP(hipEventRecord(sl2.copied, cs));
                        sl2.used = true;
                        bytes_sent += count * sizeof(Part<T>);
                        bytes_to[h] += count * sizeof(Part<T>);
                    }
                    OSP_HIP(hipEventRecord(arrive, cs));
                }
                {
                    std::lock_guard<std::mutex> l(sh->arr_m);
                    sh->arrived[(size_t)h * R + r]++;
                }
                sh->arr_cv.notify_all();
            }
            // copies queued on their streams and not complete (the host runs ahead of the device; each of them only waits
            // for its own piece to be multiplied)
            int outstanding = 0;
            for (const CopyRec &cr : copies) {
                const hipError_t q = hipEventQuery(cr.t1);
                if (q == hipErrorNotReady) outstanding++;
                (void)hipGetLastError();
            }
            max_outstanding = std::max(max_outstanding, outstanding);
        }
    });
    if (sh->failed.load()) sh->arr_cv.notify_all();
    // ---- F. merge this rank's row range, sub-panel by sub-panel, as the pieces arrive: on the merge context's stream ----
    Result *res = new Result;
    res->ctx = mctx;
    res->dtype = ops->dtype;
    res->info.M = Mh; res->info.N = N; res->info.dtype = ops->dtype;
    note_variants(mctx, res);
    float ms_merge = 0;
    guarded([&] {
        hipStream_t ms = mctx->stream;
        std::vector<const int64_t *> rps(G);
        std::vector<const void *> recs(G);
        for (int q = 0; q < G; q++) { rps[q] = sh->rp[g][q]; recs[q] = sh->recv[g][q]; }
        std::vector<uint64_t> cuts;   // local rows where a merge panel must end: the sub-panel bounds
        for (int r = 1; r < R; r++) cuts.push_back(sh->bounds[(size_t)g * R + r] - rb0);
        // before a merge panel reads the pieces: every rank's copy of the sub-panel it lies in must have been issued
        // (host side: the event exists as a recorded event) and must complete before the merge stream goes on (device side)
        std::function<void(uint64_t, uint64_t)> before = [&](uint64_t r0, uint64_t) {
            int r = 0;
            while (r + 1 < R && sh->bounds[(size_t)g * R + r + 1] - rb0 <= r0) r++;
            {
                std::unique_lock<std::mutex> l(sh->arr_m);
                sh->arr_cv.wait(l, [&] { return sh->arrived[(size_t)g * R + r] >= G || sh->failed.load(); });
            }
            if (sh->failed.load()) throw Error(OSP_ERR_HIP, "another rank failed");
            for (int q = 0; q < G; q++) OSP_HIP(hipStreamWaitEvent(ms, sh->ev[((size_t)g * G + q) * R + r], 0));
        };
        const auto t0 = std::chrono::steady_clock::now();
        osp_config_t c2 = cfg;
        c2.validate = 0;
        merge_record_parts_impl<T>(mctx, res, Mh, N, G, rps.data(), recs.data(), OSP_DEVICE, c2, &cuts, &before);
        ms_merge = ms_since(t0);
    });
    // everything this rank queued -- its copies to others included -- is complete before its buffers go away
    for (int h = 0; h < G; h++) if (h != g) (void)hipStreamSynchronize(mc->copy[g][h]);
    (void)hipStreamSynchronize(s);
    (void)hipStreamSynchronize(mctx->stream);
    const float ms_total = ms_since(t_begin);
    // how many of this rank's copies overlapped in time, and the span of its exchange (HIP event timestamps)
    int max_in_flight = 0;
    float ms_exchange = 0;
    if (t_base && !copies.empty() && !sh->failed.load()) {
        std::vector<std::pair<float, int>> edges;
        float first = 0, last = 0;
        bool any = false;
        for (const CopyRec &cr : copies) {
            float a = 0, b = 0;
            if (hipEventElapsedTime(&a, t_base, cr.t0) != hipSuccess || hipEventElapsedTime(&b, t_base, cr.t1) != hipSuccess) { (void)hipGetLastError(); continue; }
            edges.push_back({a, +1});
            edges.push_back({b, -1});
            first = any ? std::min(first, a) : a;
            last = any ? std::max(last, b) : b;
            any = true;
        }
        std::sort(edges.begin(), edges.end(), [](const std::pair<float, int> &x, const std::pair<float, int> &y) { return x.first < y.first || (x.first == y.first && x.second < y.second); });
        int cur = 0;
        for (auto &e : edges) { cur += e.second; max_in_flight = std::max(max_in_flight, cur); }
        ms_exchange = any ? last - first : 0.f;
    }
    sh->bar.wait();   // nobody releases a receive buffer another rank may still be writing
    for (void *p : pooled) ctx->release(p);
    for (int h = 0; h < G; h++)
        for (int r = 0; r < R; r++) { hipEvent_t e = sh->ev[((size_t)h * G + g) * R + r]; if (e) (void)hipEventDestroy(e); }
    for (auto &pair : slots)
        for (Slot &sl2 : pair) { if (sl2.produced) (void)hipEventDestroy(sl2.produced); if (sl2.copied) (void)hipEventDestroy(sl2.copied); }
    for (CopyRec &cr : copies) { if (cr.t0) (void)hipEventDestroy(cr.t0); if (cr.t1) (void)hipEventDestroy(cr.t1); }
    if (t_base) (void)hipEventDestroy(t_base);
    out->shard[g] = res;
    osp_multi_rank_info_t &ri = out->info.rank[g];
    ri.device = ctx->device;
    ri.k_begin = ops->k_bounds[g]; ri.k_end = ops->k_bounds[g + 1];
    ri.row_begin = rb0; ri.row_end = rb1;
    ri.partials_local = sh->P[g];
    ri.records_received = recv_total;
    ri.bytes_sent = bytes_sent;
    for (int h = 0; h < G; h++) ri.bytes_to[h] = bytes_to[h];
    ri.copy_streams = G - 1;
    ri.max_copies_outstanding = max_outstanding;
    ri.max_copies_in_flight = max_in_flight;
    ri.ms_exchange = ms_exchange;
    ri.nnz_c = res->info.nnz_c;
    ri.ms_symbolic = ms_symbolic;
    ri.ms_multiply_kernel = tm.total(PH_MUL_K);
    ri.ms_merge = ms_merge;
    ri.ms_total = ms_total;
}

template <class T>
static void multi_product(MultiContext *mc, const MultiOperands *ops, MultiResult *out, const osp_config_t &cfg) {
    const int G = (int)mc->ctx.size();
    int R = getenv("OSP_MULTI_SUBPANELS") ? atoi(getenv("OSP_MULTI_SUBPANELS")) : 4;
    R = std::max(1, std::min(R, 64));
    if (ops->M < (uint64_t)G * R) R = 1;
    MultiShared<T> sh(G, R);
    sh.row_off.assign(G, nullptr);
    sh.P.assign(G, 0);
    sh.offs.resize(G);
    sh.recv.assign(G, std::vector<Part<T> *>(G, nullptr));
    sh.rp.assign(G, std::vector<int64_t *>(G, nullptr));
    sh.ev.assign((size_t)G * G * R, nullptr);
    sh.arrived.assign((size_t)G * R, 0);
    out->shard.assign(G, nullptr);
    out->row_bounds.assign(G + 1, 0);
    out->info = osp_multi_info_t{};
    out->info.nranks = G;
    out->info.subpanels = R;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int g = 1; g < G; g++) th.emplace_back(multi_rank_main<T>, mc, ops, &sh, out, cfg, g);
    multi_rank_main<T>(mc, ops, &sh, out, cfg, 0);
    for (auto &t : th) t.join();
    (void)hipSetDevice(mc->ctx[0]->device);
    out->info.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (sh.failed.load()) throw Error(sh.err_status, sh.err_msg);
    out->info.M = ops->M; out->info.K = ops->K; out->info.N = ops->N;
    for (int g = 0; g < G; g++) {
        out->row_bounds[g] = sh.bounds[(size_t)g * R];
        out->info.nnz_c += out->shard[g]->info.nnz_c;
        out->info.partials += sh.P[g];
        out->info.bytes_exchanged += out->info.rank[g].bytes_sent;
    }
    out->row_bounds[G] = ops->M;
}

// k slabs of equal partial-product count, each uploaded to its rank
template <class T>
static void multi_upload(MultiContext *mc, MultiOperands *ops, const int64_t *acp, const uint32_t *ari, const T *av, const int64_t *brp,
                         const uint32_t *bci, const T *bv) {
    const int G = (int)mc->ctx.size();
    const uint64_t K = ops->K;
    // k slabs of equal partial-product count
    std::vector<uint64_t> cum(K + 1, 0);
    for (uint64_t k = 0; k < K; k++) cum[k + 1] = cum[k] + (uint64_t)(acp[k + 1] - acp[k]) * (uint64_t)(brp[k + 1] - brp[k]);
    ops->partials = cum[K];
    ops->k_bounds.assign(G + 1, 0);
    for (int g = 1; g < G; g++) {
        const uint64_t target = (uint64_t)((unsigned __int128)cum[K] * g / G);
        uint64_t k = std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();
        ops->k_bounds[g] = std::min(std::max(k, ops->k_bounds[g - 1]), K);
    }
    ops->k_bounds[G] = K;
    ops->slab.resize(G);
    const auto t0 = std::chrono::steady_clock::now();
    for (int g = 0; g < G; g++) {
        Context *c = mc->ctx[g];
        OSP_HIP(hipSetDevice(c->device));
        MultiOperands::Slab &sl = ops->slab[g];
        const uint64_t k0 = ops->k_bounds[g], k1 = ops->k_bounds[g + 1], Ks = k1 - k0;
        const int64_t ea = acp[k0], eb = brp[k0];
        sl.K = Ks; sl.nnz_a = (uint64_t)(acp[k1] - ea); sl.nnz_b = (uint64_t)(brp[k1] - eb);
        std::vector<int64_t> pa(Ks + 1), pb(Ks + 1);
        for (uint64_t k = 0; k <= Ks; k++) { pa[k] = acp[k0 + k] - ea; pb[k] = brp[k0 + k] - eb; }
        sl.a_colptr = (int64_t *)c->alloc((Ks + 1) * sizeof(int64_t));
        sl.b_rowptr = (int64_t *)c->alloc((Ks + 1) * sizeof(int64_t));
        sl.a_rowidx = (uint32_t *)c->alloc(std::max<uint64_t>(sl.nnz_a, 1) * sizeof(uint32_t));
        sl.b_colidx = (uint32_t *)c->alloc(std::max<uint64_t>(sl.nnz_b, 1) * sizeof(uint32_t));
        sl.a_vals = c->alloc(std::max<uint64_t>(sl.nnz_a, 1) * sizeof(T));
        sl.b_vals = c->alloc(std::max<uint64_t>(sl.nnz_b, 1) * sizeof(T));
        copy_h2d(sl.a_colptr, pa.data(), (Ks + 1) * sizeof(int64_t), c->stream);
        copy_h2d(sl.b_rowptr, pb.data(), (Ks + 1) * sizeof(int64_t), c->stream);
        copy_h2d(sl.a_rowidx, ari + ea, sl.nnz_a * sizeof(uint32_t), c->stream);
        copy_h2d(sl.b_colidx, bci + eb, sl.nnz_b * sizeof(uint32_t), c->stream);
        copy_h2d(sl.a_vals, av + ea, sl.nnz_a * sizeof(T), c->stream);
        copy_h2d(sl.b_vals, bv + eb, sl.nnz_b * sizeof(T), c->stream);
    }
    ops->ms_upload = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}


}  // namespace osp
