// tools/bench_copy.hip -- which plain copy reaches the highest HBM rate on this box (the measured roof of bench.py).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bench_copy.hip -o tools/bench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// grid-stride, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_stride(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const uint64_t j = i + u * stride; if (j < n) v[u] = NT ? __builtin_nontemporal_load(&src[j]) : src[j]; }
#pragma unroll
        for (int u = 0; u < U; u++) { const uint64_t j = i + u * stride; if (j < n) { if (NT) __builtin_nontemporal_store(v[u], &dst[j]); else dst[j] = v[u]; } }
    }
}
// every workgroup copies one contiguous chunk
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, uint64_t n) {
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x, b = blockIdx.x * per, e = b + per < n ? b + per : n;
    for (uint64_t i = b + threadIdx.x; i < e; i += U * 256) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const uint64_t j = i + u * 256; if (j < e) v[u] = NT ? __builtin_nontemporal_load(&src[j]) : src[j]; }
#pragma unroll
        for (int u = 0; u < U; u++) { const uint64_t j = i + u * 256; if (j < e) { if (NT) __builtin_nontemporal_store(v[u], &dst[j]); else dst[j] = v[u]; } }
    }
}
template <class F> double timeit(F f, uint64_t bytes, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) f();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
}
int main(int argc, char **argv) {
    const uint64_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : (2ull << 30);
    const uint64_t n = bytes / 16;
    u32x4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMemset(src, 0x5a, bytes));
    for (unsigned g : {1024u, 2048u, 4096u, 8192u, 16384u, 65536u}) {
        printf("grid %6u: stride U1 %7.0f  U4 %7.0f  U4nt %7.0f  U8 %7.0f | chunk U4 %7.0f  U4nt %7.0f  U8 %7.0f GB/s\n", g,
               timeit([&] { copy_stride<1, false><<<g, 256>>>(src, dst, n); }, bytes), timeit([&] { copy_stride<4, false><<<g, 256>>>(src, dst, n); }, bytes),
               timeit([&] { copy_stride<4, true><<<g, 256>>>(src, dst, n); }, bytes), timeit([&] { copy_stride<8, false><<<g, 256>>>(src, dst, n); }, bytes),
               timeit([&] { copy_chunk<4, false><<<g, 256>>>(src, dst, n); }, bytes), timeit([&] { copy_chunk<4, true><<<g, 256>>>(src, dst, n); }, bytes),
               timeit([&] { copy_chunk<8, false><<<g, 256>>>(src, dst, n); }, bytes));
    }
    {
        const unsigned g = (unsigned)((n + 255) / 256);
        printf("one lane per 16 B (grid %u): %7.0f GB/s; 4 per lane: %7.0f\n", g, timeit([&] { copy_stride<1, false><<<g, 256>>>(src, dst, n); }, bytes),
               timeit([&] { copy_stride<4, false><<<(g + 3) / 4, 256>>>(src, dst, n); }, bytes));
    }
    printf("hipMemcpyAsync D2D: %7.0f GB/s\n", timeit([&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); }, bytes));
    printf("(read-only and write-only for orientation)\n");
    return 0;
}
