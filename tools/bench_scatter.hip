// tools/bench_scatter.hip -- how fast can 16-entry chunks be written to random places of a big buffer?
// SoA (u32 col | f64 val arrays) vs AoS (12-byte records), aligned vs misaligned chunk starts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct __attribute__((packed, aligned(4))) Rec { uint32_t c; double v; };

__device__ inline uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }
// chunk id c (of nch) -> destination slot: a bijection on [0, nch) (nch power of two): odd multiply + xor
// with a window: chunks are written window by window (window = wch chunks, power of two), random inside
__device__ uint64_t g_wch;
__device__ inline uint64_t slot_of(uint64_t c, uint64_t nch) {
    const uint64_t wch = g_wch;
    const uint64_t win = c & ~(wch - 1);
    return win | (((c * 0x9E3779B1ull) ^ 0x5bd1e995ull) & (wch - 1));
}

// MODE 2: the same AoS bytes, written as single dwords (lane -> dword of the chunk's byte range): 3 store
// instructions of one dword per lane instead of one of three dwords per lane
__global__ void scatter_dwords_kernel(Rec *rec, uint64_t nch, int clen, int shift) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 64 / clen;
    const uint64_t wave = gid >> 6; const int lane = gid & 63;
    uint32_t *out = (uint32_t *)rec;
    for (int it = 0; it < 8; it++) {
        const uint64_t c0 = (wave * 8 + it) * per;
        for (int q = 0; q < 3; q++) {
            const int g = q * 64 + lane;
            const int jj = g / (3 * clen), w = g % (3 * clen);
            const uint64_t c = c0 + jj;
            if (c >= nch) continue;
            const uint64_t dst = (slot_of(c, nch) * clen + shift) * 3 + w;   // dword index
            const int l = w / 3, comp = w % 3;
            const double v = (double)l;
            const uint32_t word = comp == 0 ? (uint32_t)c : comp == 1 ? (uint32_t)__double_as_longlong(v) : (uint32_t)(__double_as_longlong(v) >> 32);
            out[dst] = word;
        }
    }
}

template <int MODE>  // 0 SoA, 1 AoS
__global__ void scatter_kernel(uint32_t *pcol, double *pval, Rec *rec, uint64_t nch, int clen, int shift) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 64 / clen;                       // chunks per wave instruction
    const uint64_t wave = gid >> 6; const int lane = gid & 63;
    const int jj = lane / clen, l = lane % clen;
    if (jj >= per) return;
    for (int it = 0; it < 8; it++) {
        const uint64_t c = (wave * 8 + it) * per + jj;
        if (c >= nch) return;
        const uint64_t dst = slot_of(c, nch) * clen + shift + l;
        if (MODE == 0) { pcol[dst] = (uint32_t)c; pval[dst] = (double)l; }
        else { Rec r; r.c = (uint32_t)c; r.v = (double)l; rec[dst] = r; }
    }
}
int main(int argc, char **argv) {
    const int clen = argc > 1 ? atoi(argv[1]) : 16;
    const uint64_t nch = 1ull << 26;  // 67M chunks
    const uint64_t P = nch * clen + 64;
    uint32_t *pcol; double *pval; Rec *rec;
    CK(hipMalloc(&pcol, P * 4)); CK(hipMalloc(&pval, P * 8)); CK(hipMalloc(&rec, P * 12));
    const int per = 64 / clen;
    const uint64_t waves = (nch + per * 8 - 1) / (per * 8);
    const unsigned grid = (unsigned)((waves * 64 + 255) / 256);
    for (uint64_t wmb : {16384ull, 512ull, 128ull, 32ull}) {
    uint64_t wch = 1; while (wch * 2 * clen * 12 <= wmb * 1024 * 1024 && wch * 2 <= nch) wch *= 2;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wch), &wch, 8));
    printf("-- window %llu chunks = %.0f MB of staging\n", (unsigned long long)wch, wch * clen * 12.0 / 1048576);
    for (int shift : {0, 5}) for (int mode : {0, 1, 2}) {
        std::vector<float> t;
        for (int r = 0; r < 4; r++) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, 0));
            if (mode == 0) scatter_kernel<0><<<grid, 256>>>(pcol, pval, rec, nch, clen, shift);
            else if (mode == 1) scatter_kernel<1><<<grid, 256>>>(pcol, pval, rec, nch, clen, shift);
            else scatter_dwords_kernel<<<grid, 256>>>(rec, nch, clen, shift);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("clen %d shift %d %s: %.3f ms  %.1f GB/s\n", clen, shift, mode == 0 ? "SoA" : mode == 1 ? "AoS x3" : "AoS dwords", t[1], 12.0 * nch * clen / (t[1] * 1e-3) / 1e9);
    }
    }
    return 0;
}
