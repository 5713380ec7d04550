// osp_context.h -- the library's context (device, streams, buffer pool), scratch allocations, phase timers, host <-> device
// copies and read-backs.  Part of the ONE translation unit osp_api.hip (included there, inside namespace osp); split out of it in round 5.
#pragma once

// ---- context: device, stream, buffer pool ------------------------------------------------------
struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream of a product with several panels: the plan of panel p+1 runs on it beside the multiply of panel p
    // (merge_pipeline); created on first use, fork/join by the two events
    hipStream_t aux = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    void need_aux() {
        if (aux) return;
        if (hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) != hipSuccess) { aux = nullptr; throw Error(OSP_ERR_HIP, "hipStreamCreate failed"); }
        if (hipEventCreateWithFlags(&aux_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&aux_join, hipEventDisableTiming) != hipSuccess) {
            drop_aux();   // (never a stream without its events: the next product would skip the creation)
            throw Error(OSP_ERR_HIP, "hipEventCreate failed");
        }
    }
    // The pool is not stream-aware: a released block goes to whoever asks next.  While the plan of the next panel runs on the
    // second stream that is safe only as long as NOTHING is released between the fork event and the plan's own allocations
    // (a block the multiply still reads would go straight to a plan kernel).  The window is marked and every release inside
    // it counted: merge_pipeline turns a non-zero count into an error instead of a silent corruption.
    bool fork_window = false;
    uint64_t releases_in_fork_window = 0;
    void drop_aux() {
        if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); aux = nullptr; }
        if (aux_fork) { (void)hipEventDestroy(aux_fork); aux_fork = nullptr; }
        if (aux_join) { (void)hipEventDestroy(aux_join); aux_join = nullptr; }
    }
    std::multimap<size_t, void *> free_list;
    std::map<void *, size_t> live;
    size_t pooled_bytes = 0;
    uint32_t cus = 256;  // persistent kernels size their grids from this
    // stable radix ranks from the return order of LDS atomics (true) or from ballot matching (false): decided when the
    // context is created (self-test; OSP_RANK=ballot|atomic overrides), see osp_prims.h
    bool rank_atomic = OSP_RANK_ATOMIC != 0;
    // dense accumulation of narrow over-long segments (osp_split.h) by LDS floating-point atomics (f64) or by ballot ranks
    // and rounds: same decision procedure (self-test; OSP_DENSE_ADD=ballot|atomic overrides)
    bool dense_atomic[2] = {true, true};  // [0] f32, [1] f64
    // pool misses (OSP_VERBOSE prints them per product): device allocations are slow, a product should not need any
    // once the pool is warm
    Context *sibling = nullptr;   // the other context of the same rank and device (osp_multi.h), same host thread: its pooled blocks are freed before an allocation fails
    uint64_t malloc_calls = 0, malloc_bytes = 0;
    double malloc_ms = 0;

    static size_t bucket(size_t bytes) {
        if (bytes < 4096) return 4096;
        size_t p = 1;
        while (p * 2 <= bytes) p *= 2;
        size_t step = p / 8;
        return (bytes + step - 1) / step * step;
    }
    // Debugging aid: OSP_GUARD=1 gives every buffer its own allocation with 4 KiB of 0xA5 before it and from the
    // end of the REQUESTED size to the end of the allocation, and checks both zones when the buffer is released --
    // a kernel that writes a little past (or before) its buffer is named instead of corrupting a neighbour.
    // (Bucket rounding normally hides such writes unless the request happens to fill its bucket.)
    static constexpr size_t kGuard = 4096;
    struct GuardRec { char *base; size_t total, bytes; };
    std::map<void *, GuardRec> guarded;
    static bool guard_mode() { static const bool g = getenv("OSP_GUARD") != nullptr; return g; }
    // OSP_GUARD=2 ("electric fence"): every buffer is mapped through the virtual-memory API so that it ENDS at the end
    // of its mapping, with the address range behind it left unmapped -- an access past the end of a buffer, READS
    // included, faults on the spot.  Nothing is ever unmapped or reused in this mode (small test inputs only): early
    // experiments that did unmap showed stale translations, which look like bugs and are not.
    static bool fence_mode() { static const bool g = getenv("OSP_GUARD") && atoi(getenv("OSP_GUARD")) == 2; return g; }
    void *alloc_fenced(size_t bytes) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        OSP_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
        const size_t map_size = (bytes + gran - 1) / gran * gran;
        char *va = nullptr;
        hipMemGenericAllocationHandle_t h;
        OSP_HIP(hipMemAddressReserve((void **)&va, map_size + gran, gran, nullptr, 0));
        OSP_HIP(hipMemCreate(&h, map_size, &prop, 0));
        OSP_HIP(hipMemMap(va, map_size, 0, h, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        OSP_HIP(hipMemSetAccess(va, map_size, &acc, 1));
        char *user = va + (map_size - bytes) / 16 * 16;
        OSP_HIP(hipMemsetAsync(va, 0xA5, map_size, stream));
        if (getenv("OSP_VERBOSE")) fprintf(stderr, "[osp] fence: %zu bytes at [%p, %p), mapping ends at %p\n", bytes, (void *)user, (void *)(user + bytes), (void *)(va + map_size));
        return user;
    }
    void *alloc_guarded(size_t bytes) {
        const size_t total = bucket(bytes + 2 * kGuard);
        char *base = nullptr;
        hipError_t e = hipMalloc((void **)&base, total);
        if (e != hipSuccess) { (void)hipGetLastError(); throw Error(OSP_ERR_ALLOC, "hipMalloc of " + std::to_string(total) + " bytes failed (guard mode)"); }
        (void)hipMemsetAsync(base, 0xA5, kGuard, stream);
        (void)hipMemsetAsync(base + kGuard + bytes, 0xA5, total - kGuard - bytes, stream);
        guarded[base + kGuard] = GuardRec{base, total, bytes};
        return base + kGuard;
    }
    void release_guarded(void *p) {
        auto it = guarded.find(p);
        if (it == guarded.end()) return;
        const GuardRec g = it->second;
        guarded.erase(it);
        (void)hipStreamSynchronize(stream);
        const size_t tail = std::min<size_t>(g.total - kGuard - g.bytes, 1 << 20);
        std::vector<unsigned char> h(kGuard + tail);
        (void)hipMemcpy(h.data(), g.base, kGuard, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h.data() + kGuard, g.base + kGuard + g.bytes, tail, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < h.size(); i++) {
            if (h[i] != 0xA5) {
                const long long off = i < kGuard ? (long long)i - (long long)kGuard : (long long)(i - kGuard);
                fprintf(stderr, "[osp] OSP_GUARD: buffer of %zu bytes was written %s it: first damaged byte at %s%lld (value 0x%02x)\n",
                        g.bytes, i < kGuard ? "BEFORE" : "PAST the end of", i < kGuard ? "offset " : "end+", off, h[i]);
                fflush(stderr);
                abort();
            }
        }
        (void)hipFree(g.base);
    }
    void *alloc(size_t bytes) {
        if (fence_mode()) return alloc_fenced(bytes ? bytes : 1);
        if (guard_mode()) return alloc_guarded(bytes ? bytes : 1);
        size_t b = bucket(bytes ? bytes : 1);
        // best fit among pooled blocks: anything from b to 1.5 b is reused (buffer sizes drift from panel to
        // panel and from call to call; hipMalloc / hipFree of multi-GB blocks cost far more than the slack)
        auto it = free_list.lower_bound(b);
        void *p = nullptr;
        if (it != free_list.end() && it->first <= b + b / 2) {
            p = it->second;
            b = it->first;
            free_list.erase(it);
            pooled_bytes -= b;
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMalloc(&p, b);
            if (e != hipSuccess) {
                // out of memory.  First choice: a pooled block that is merely too generous for the 1.5x rule (the
                // multi-GB scratch of a panel drifts from panel to panel; freeing such blocks only to allocate them
                // again cost a third of the run time of the streamed Graph500 products).
                (void)hipGetLastError();
                it = free_list.lower_bound(b);
                if (it != free_list.end()) {
                    p = it->second;
                    b = it->first;
                    free_list.erase(it);
                    pooled_bytes -= b;
                    live[p] = b;
                    return p;
                }
            }
            while (e != hipSuccess && !free_list.empty()) {
                // still nothing: give the largest pooled blocks back until the request fits
                (void)hipGetLastError();
                auto big = std::prev(free_list.end());
                (void)hipFree(big->second);
                pooled_bytes -= big->first;
                free_list.erase(big);
                e = hipMalloc(&p, b);
            }
            // ... then the sibling's (a rank of the multi-GPU path has two contexts on one device, used by one host thread: what the
            // multiply's context has released is no use to anybody while the merge's context runs out of memory)
            while (e != hipSuccess && sibling && !sibling->free_list.empty()) {
                (void)hipGetLastError();
                auto big = std::prev(sibling->free_list.end());
                (void)hipFree(big->second);
                sibling->pooled_bytes -= big->first;
                sibling->free_list.erase(big);
                e = hipMalloc(&p, b);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                throw Error(OSP_ERR_ALLOC, "hipMalloc of " + std::to_string(b) + " bytes failed");
            }
            malloc_calls++;
            malloc_bytes += b;
            const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            malloc_ms += dt;
            if (b >= (1ull << 30) && getenv("OSP_VERBOSE"))
                fprintf(stderr, "[osp]   pool miss: hipMalloc of %.2f GB took %.0f ms (pooled %.1f GB in %zu blocks, live %zu blocks)\n", b / 1e9, dt,
                        pooled_bytes / 1e9, free_list.size(), live.size());
        }
        live[p] = b;
        // debugging aid: OSP_POISON=1 fills every buffer with 0xFF bytes, so that a read of memory nobody
        // wrote fails the same way on every run instead of depending on what the pool hands back
        static const bool poison = getenv("OSP_POISON") != nullptr;
        if (poison) (void)hipMemsetAsync(p, 0xff, b, stream);
        return p;
    }
    void release(void *p) {
        if (!p) return;
        if (fork_window) releases_in_fork_window++;
        if (fence_mode()) return;  // leaked on purpose, see fence_mode()
        if (guard_mode()) { release_guarded(p); return; }
        auto it = live.find(p);
        if (it == live.end()) return;
        free_list.emplace(it->second, p);
        pooled_bytes += it->second;
        live.erase(it);
    }
    // Leave `bytes` of device memory to others (the consumer of a streamed panel runs its own kernels and allocations
    // while this pool may hold everything): hand pooled blocks back, small ones first -- they are the cheap ones to
    // allocate again.
    void ensure_free(size_t bytes) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
        while (free_b < bytes && !free_list.empty()) {
            auto it = free_list.begin();
            (void)hipFree(it->second);
            pooled_bytes -= it->first;
            free_b += it->first;
            free_list.erase(it);
        }
    }
    void trim() {
        for (auto &kv : free_list) (void)hipFree(kv.second);
        free_list.clear();
        pooled_bytes = 0;
    }
};

// RAII scratch that returns to the pool
struct Scratch {
    Context *ctx;
    std::vector<void *> ptrs;
    explicit Scratch(Context *c) : ctx(c) {}
    ~Scratch() { for (void *p : ptrs) ctx->release(p); }
    template <class T> T *get(uint64_t n) {
        void *p = ctx->alloc((size_t)(n ? n : 1) * sizeof(T));
        ptrs.push_back(p);
        return (T *)p;
    }
    void drop(void *p) {
        for (auto &q : ptrs) if (q == p) { ctx->release(p); q = nullptr; }
    }
};

struct Result {
    Context *ctx = nullptr;
    int dtype = OSP_F64;
    osp_result_info_t info{};
    int64_t *rowptr = nullptr;
    uint32_t *colidx = nullptr;
    void *vals = nullptr;
    bool partials = false;  // osp_spgemm_partials: rowptr = record offsets per row, vals = the packed records, no colidx
};

struct PhaseTimer {
    hipStream_t s;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[10];
    explicit PhaseTimer(hipStream_t st) : s(st) {}
    ~PhaseTimer() {
        for (auto &v : ev) for (auto &p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    }
    hipStream_t on[10] = {};   // the stream the open interval of a phase was begun on (a panel's plan may run on the second stream)
    void begin(int ph, hipStream_t st = nullptr) {
        hipEvent_t a, b;
        OSP_HIP(hipEventCreate(&a));
        OSP_HIP(hipEventCreate(&b));
        on[ph] = st ? st : s;
        OSP_HIP(hipEventRecord(a, on[ph]));
        ev[ph].push_back({a, b});
    }
    void end(int ph) { OSP_HIP(hipEventRecord(ev[ph].back().second, on[ph])); }
    float total(int ph) {
        float t = 0;
        for (auto &p : ev[ph]) { float ms = 0; if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) t += ms; }
        return t;
    }
};
// start / stop events of one call, released on every exit path
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    EventPair() {
        OSP_HIP(hipEventCreate(&a));
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); throw Error(OSP_ERR_HIP, "hipEventCreate failed"); }
    }
    ~EventPair() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    EventPair(const EventPair &) = delete;
    EventPair &operator=(const EventPair &) = delete;
    float ms() const { float t = 0; (void)hipEventElapsedTime(&t, a, b); return t; }
};
enum { PH_SYM = 0, PH_MUL = 1, PH_MERGE = 2, PH_COMPACT = 3, PH_MUL_K = 4, PH_MERGE_K = 5, PH_SPLIT_K = 6, PH_PLAN_K = 7, PH_HUB_K = 8, PH_EXPAND_K = 9 };

// debugging aid: OSP_SYNC=1 waits for the stream at the marked points of a product and names them on stderr, so
// that an asynchronous GPU fault is pinned to the step that caused it (the last name printed COMPLETED)
#ifdef OSP_CHECK_DESC
static unsigned long long *g_crumbs_host = nullptr;
static void crumbs_init() {
    if (g_crumbs_host) return;
    if (hipHostMalloc((void **)&g_crumbs_host, 4096 * 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return;
    memset(g_crumbs_host, 0, 4096 * 8 * sizeof(unsigned long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(osp_crumbs), &g_crumbs_host, sizeof(g_crumbs_host));
}
static void crumbs_dump(const char *what) {
    if (!g_crumbs_host) return;
    int shown = 0, hist[8] = {0};
    for (int b = 0; b < 4096; b++) hist[g_crumbs_host[(size_t)b * 8] & 7u]++;
    fprintf(stderr, "[osp] crumb phases: none %d, tile start %d, reload %d, output %d, tile done %d, kernel start %d, kernel end %d\n", hist[0], hist[1], hist[2], hist[3],
            hist[4], hist[5], hist[7]);
    for (int b = 0; b < 4096 && shown < 24; b++) {
        const unsigned long long *p = g_crumbs_host + (size_t)b * 8;
        const unsigned ph = (unsigned)(p[0] & 255u);
        if (ph == 0 || ph == 7) continue;
        fprintf(stderr, "[osp] crumb block %d: phase %u NT %llu ABL %llu tile %llu s %llu n %llu ra %llu | %llu %llu %llu\n", b, ph, (p[0] >> 8) & 0xffffu, p[0] >> 32,
                p[1], p[2], p[3], p[7], p[4], p[5], p[6]);
        shown++;
    }
    fprintf(stderr, "[osp] (%s: crumbs of workgroups that were inside a tile)\n", what);
}
#endif
static inline void dbg_sync(hipStream_t s, const char *what) {
    static const bool on = getenv("OSP_SYNC") != nullptr;
    if (!on) return;
    const hipError_t e1 = hipStreamSynchronize(s), e2 = hipGetLastError();
#ifdef OSP_CHECK_DESC
    if (e1 != hipSuccess || e2 != hipSuccess) crumbs_dump(what);
    else if (g_crumbs_host) memset(g_crumbs_host, 0, 4096 * 8 * sizeof(unsigned long long));
#endif
#ifdef OSP_CHECK_DESC
    {
        unsigned long long h[16] = {0};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(osp_desc_bad), sizeof(h)) == hipSuccess && h[0]) {
            fprintf(stderr, "[osp] DESCRIPTOR MISMATCH before '%s': %llu threads; first: tile %llu tid %llu s(lds) %llu s(mem) %llu n(lds) %llu n(mem) %llu lvl(lds) %llu block %llu NT %llu\n",
                    what, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
            unsigned long long z[16] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(osp_desc_bad), z, sizeof(z));
        }
    }
#endif
    if (e1 != hipSuccess || e2 != hipSuccess) {
        fprintf(stderr, "[osp] FAILED in: %s (%s)\n", what, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        fflush(stderr);
        throw Error(OSP_ERR_HIP, std::string("device error in phase: ") + what);
    }
    fprintf(stderr, "[osp] ok: %s\n", what);
    fflush(stderr);
}
static inline unsigned grid_for(uint64_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
static inline int bits_for(uint64_t n) {  // bits needed to represent values in [0, n)
    int b = 0;
    while (b < 64 && (n > (1ull << b))) b++;
    return b;
}

// Host <-> device copies go through two pinned staging buffers owned by this library (the memcpy into one overlaps the
// DMA out of the other) rather than straight from or to the caller's pageable memory.  Kept for what it guarantees, not
// for a fault it avoids (round 1's intermittent GPU fault had another cause): the caller's pages are
// never pinned or unpinned behind its back, the copy is complete when the call returns, and the pinned footprint is two
// 16 MB chunks whatever the operand size.  Device-resident operands (bench.py, the multi-GPU path) never come here.
constexpr size_t kStageChunk = 16u << 20;
struct Pinned {
    char *p = nullptr;     // two halves of `half` bytes each
    size_t half = 0;
    hipEvent_t done[2] = {nullptr, nullptr};  // the DMA out of / into half i has finished
    uint64_t *collect = nullptr;              // device words a read-back of several scalars is gathered into (Gather)
    // (never freed: a thread_local destructor can run after the HIP runtime has shut down)
    void reserve(size_t want) {
        want = std::min(std::max<size_t>(want, 64), kStageChunk);
        if (!done[0]) { OSP_HIP(hipEventCreateWithFlags(&done[0], hipEventDisableTiming)); OSP_HIP(hipEventCreateWithFlags(&done[1], hipEventDisableTiming)); }
        if (half >= want) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        half = 0;
        if (hipHostMalloc((void **)&p, 2 * want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            throw Error(OSP_ERR_ALLOC, "hipHostMalloc of the staging buffer failed");
        }
        half = want;
    }
};
// One staging object per (thread, device): its events belong to the device that was current when they were created,
// and recording them on another device's stream fails with "invalid resource handle" -- a thread may own contexts on
// several devices (the multi-GPU entry point does).
static Pinned &pinned_buffer() {
    static thread_local std::map<int, Pinned> per_device;  // lives as long as the thread; at most 32 MB per device
    int dev = 0;
    OSP_HIP(hipGetDevice(&dev));
    return per_device[dev];
}
static void copy_h2d(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    Pinned &pb = pinned_buffer();
    pb.reserve(bytes);
    const size_t chunk = pb.half;
    int i = 0;
    size_t nchunks = 0;
    for (size_t off = 0; off < bytes; off += chunk, i ^= 1, nchunks++) {
        const size_t n = std::min(chunk, bytes - off);
        if (nchunks >= 2) OSP_HIP(hipEventSynchronize(pb.done[i]));  // the DMA that last read this half
        memcpy(pb.p + i * chunk, (const char *)src + off, n);
        OSP_HIP(hipMemcpyAsync((char *)dst + off, pb.p + i * chunk, n, hipMemcpyHostToDevice, s));
        OSP_HIP(hipEventRecord(pb.done[i], s));
    }
    OSP_HIP(hipStreamSynchronize(s));  // complete on return: the staging halves are free again
}
static void copy_d2h(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    Pinned &pb = pinned_buffer();
    pb.reserve(bytes);
    const size_t chunk = pb.half;
    // DMA of chunk j+1 runs while chunk j is copied out of its half
    size_t off_prev = 0, n_prev = 0;
    int i = 0;
    bool have_prev = false;
    for (size_t off = 0; off < bytes; off += chunk, i ^= 1) {
        const size_t n = std::min(chunk, bytes - off);
        OSP_HIP(hipMemcpyAsync(pb.p + i * chunk, (const char *)src + off, n, hipMemcpyDeviceToHost, s));
        OSP_HIP(hipEventRecord(pb.done[i], s));
        if (have_prev) {
            OSP_HIP(hipEventSynchronize(pb.done[i ^ 1]));
            memcpy((char *)dst + off_prev, pb.p + (i ^ 1) * chunk, n_prev);
        }
        off_prev = off; n_prev = n; have_prev = true;
    }
    OSP_HIP(hipEventSynchronize(pb.done[i ^ 1]));
    memcpy((char *)dst + off_prev, pb.p + (i ^ 1) * chunk, n_prev);
}
// Zeroing up to four small arrays with ONE kernel.  hipMemsetAsync is a blit with barriers around it: in a kernel trace
// each one costs 2-8 us plus ~10 us of idle stream before the next kernel starts, and a product issues half a dozen
// (counters, flags, the tile status words); kernels queued behind kernels start without a gap.
struct ZeroRegions {
    uint32_t *p[4];
    uint64_t words[4];
    uint32_t val[4];   // the word every element of the region is set to (zero_async: 0; fill_async: e.g. 0xffffffff)
};
__global__ void zero_regions_kernel(const ZeroRegions z) {
#pragma unroll
    for (int r = 0; r < 4; r++)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.words[r]; i += (uint64_t)gridDim.x * blockDim.x) z.p[r][i] = z.val[r];
}
struct FillRegion { void *p; size_t bytes; uint32_t word; };
static void fill_async(hipStream_t s, std::initializer_list<FillRegion> regions) {   // (pointer, bytes: multiples of 4, the word)
    ZeroRegions z{};
    int n = 0;
    uint64_t most = 0;
    for (auto &r : regions) {
        if (n == 4 || (r.bytes & 3) || ((uintptr_t)r.p & 3)) throw Error(OSP_ERR_ARG, "fill_async: at most four word-aligned regions");
        z.p[n] = (uint32_t *)r.p;
        z.words[n] = r.bytes / 4;
        z.val[n] = r.word;
        most = std::max<uint64_t>(most, z.words[n]);
        n++;
    }
    if (most == 0) return;
    zero_regions_kernel<<<(unsigned)std::min<uint64_t>((most + 255) / 256, 2048), 256, 0, s>>>(z);
}
static void zero_async(hipStream_t s, std::initializer_list<std::pair<void *, size_t>> regions) {   // (pointer, bytes: multiples of 4)
    ZeroRegions z{};
    int n = 0;
    uint64_t most = 0;
    for (auto &r : regions) {
        if (n == 4 || (r.second & 3) || ((uintptr_t)r.first & 3)) throw Error(OSP_ERR_ARG, "zero_async: at most four word-aligned regions");
        z.p[n] = (uint32_t *)r.first;
        z.words[n] = r.second / 4;
        most = std::max<uint64_t>(most, z.words[n]);
        n++;
    }
    if (most == 0) return;
    zero_regions_kernel<<<(unsigned)std::min<uint64_t>((most + 255) / 256, 2048), 256, 0, s>>>(z);
}
// several device scalars with ONE wait: a blocking read-back is a stream round trip, and a small product makes a dozen.
// Three or more values are first gathered into consecutive device words by one tiny kernel and come back in ONE copy (a
// copy of 8 bytes occupies the stream for 5-8 us: ten of them cost what the gather and its copy cost four times over).
constexpr int kGatherMax = 24;
struct GatherSrcs {
    const void *p[kGatherMax];
    uint8_t bytes[kGatherMax];
};
__global__ void gather_scalars_kernel(const GatherSrcs g, int n, uint64_t *out) {
    const int i = threadIdx.x;
    if (i < n) out[i] = g.bytes[i] == 8 ? *static_cast<const uint64_t *>(g.p[i]) : (uint64_t) * static_cast<const uint32_t *>(g.p[i]);
}
struct Gather {
    hipStream_t s;
    Pinned &pb;
    GatherSrcs srcs;
    void *dst[kGatherMax];
    int n = 0;
    explicit Gather(hipStream_t st) : s(st), pb(pinned_buffer()) { pb.reserve(4096); }
    template <class T> void add(T *host_dst, const T *dptr) {
        static_assert(sizeof(T) == 4 || sizeof(T) == 8, "read-backs are 32- or 64-bit scalars");
        if (n == kGatherMax) throw Error(OSP_ERR_ARG, "too many values in one read-back");
        srcs.p[n] = dptr;
        srcs.bytes[n] = (uint8_t)sizeof(T);
        dst[n++] = host_dst;
    }
    void wait() {
        if (n == 0) return;
        uint64_t *pin = reinterpret_cast<uint64_t *>(pb.p);
        if (n >= 3) {
            if (!pb.collect) OSP_HIP(hipMalloc((void **)&pb.collect, kGatherMax * sizeof(uint64_t)));
            gather_scalars_kernel<<<1, kWave, 0, s>>>(srcs, n, pb.collect);
            OSP_HIP(hipMemcpyAsync(pin, pb.collect, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        } else {
            for (int i = 0; i < n; i++) OSP_HIP(hipMemcpyAsync(pin + i, srcs.p[i], srcs.bytes[i], hipMemcpyDeviceToHost, s));
        }
        OSP_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < n; i++) memcpy(dst[i], pin + i, srcs.bytes[i]);   // (little endian: the low bytes of the word)
        n = 0;
    }
};
template <class T> static T d2h(const T *dptr, hipStream_t s) {
    T v;
    copy_d2h(&v, dptr, sizeof(T), s);
    return v;
}

