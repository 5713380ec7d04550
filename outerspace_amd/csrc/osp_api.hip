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
};
__global__ void zero_regions_kernel(const ZeroRegions z) {
#pragma unroll
    for (int r = 0; r < 4; r++)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.words[r]; i += (uint64_t)gridDim.x * blockDim.x) z.p[r][i] = 0u;
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
    const int direct_fine = getenv("OSP_DIRECT_FINE") ? std::max(0, std::min(atoi(getenv("OSP_DIRECT_FINE")), 4)) : 2;
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
            OSP_HIP(hipMemsetAsync(pl.vrun_off, 0xff, (pl.nvirt + 1) * sizeof(uint32_t), s));   // kNoRuns: segments of rows that are not gathered
            gp.rdbase = rdbase; gp.vrun_off = pl.vrun_off; gp.vrun_end = pl.vrun_end;
            gp.rowruns = sc.get<uint32_t>(nlong);
            OSP_HIP(hipMemsetAsync(gp.rowruns, 0xff, (uint64_t)nlong * sizeof(uint32_t), s));
            pl.nwritten = gp.nwritten = sc.get<uint32_t>(1);
            zero_async(s, {{pl.nwritten, sizeof(uint32_t)}});
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
                            "prefixes + output %.1f %%\n", 100 * hp[0] / tot, 100 * hp[1] / tot, 100 * hp[2] / tot, 100 * hp[3] / tot, 100 * hp[4] / tot,
                    100 * hp[5] / tot);
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
            if (pl.ga.runs)
                expand_segments_kernel<T><<<nseg_long, kExpandThreads, 0, s>>>(p1.long_rows, nseg_long, vrow_off, pl.vrun_off, pl.vrun_end, pl.ga.runs,
                                                                              pl.ga.b_colidx, pl.ga.b_vals, qstage);
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
    free_b += ctx->pooled_bytes;
    uint64_t cap = cap_cfg;
    // streaming: the panel's output buffer (at most one record per partial product) comes out of the same budget
    // (debugging aid: OSP_STAGE_FACTOR overrides the number of record sizes budgeted per staged partial product)
    // what a staged partial product needs: its record in the staging buffer, for 9 of 10 another one in the second buffer,
    // and a few per cent for tile tables and the cells of the direct rows -- 2.0 record sizes; 2.6 budgets 30 % on top of
    // that (3.3 until round 3: R-MAT-22 mild ran as 4 panels, now 3: one panel's planning, launches and read-backs less)
    // (round 5, gathered rows: nothing is written for nine records of ten, but both buffers are still addressed by the rows'
    // positions -- allocated in full -- and the run table is sized by a bound: measured 2.25 record sizes per product; 2.5)
    const double per_record = getenv("OSP_STAGE_FACTOR") ? atof(getenv("OSP_STAGE_FACTOR")) : (sink ? 3.7 : (ds && ds->gather) ? 2.5 : 2.6);
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
