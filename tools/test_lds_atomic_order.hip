// Does one LDS atomic-add instruction hand out its old values in ascending lane order among the lanes that hit the same
// address?  (Undocumented; a stable radix rank could use it.)  Compares against the ballot-based match for random digits of
// several widths, including packed 16-bit counters (two digits per 32-bit word).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I outerspace_amd/csrc tools/test_lds_atomic_order.hip -o tools/test_lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "osp_kernels.h"
using namespace osp;

__global__ __launch_bounds__(256) void order_kernel(int bits, int rounds, uint64_t seed, unsigned long long *bad, unsigned long long *checked) {
    __shared__ uint32_t cnt[4][512];  // packed: digit d -> half (d & 1) of word d >> 1
    const unsigned tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    uint64_t x = seed + (uint64_t)blockIdx.x * 1315423911ull + tid * 2654435761ull;
    unsigned long long mybad = 0, mychk = 0;
    for (int r = 0; r < rounds; r++) {
        for (int i = lane; i < 512; i += 64) cnt[w][i] = 0;
        for (int it = 0; it < 6; it++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            // skewed sometimes: half of the rounds draw from a narrow range to force many conflicts
            unsigned dg = (unsigned)(x >> 20) & ((1u << bits) - 1u);
            if (r & 1) dg &= 7u;
            const bool valid = ((x >> 50) & 15u) != 0;  // some lanes sit out
            unsigned rk, c;
            wave_match_digit<10>(dg, valid, rk, c);
            // expected: counter before this instruction + rank among equal lanes
            const uint32_t before = (cnt[w][dg >> 1] >> (16 * (dg & 1))) & 0xffffu;
            __builtin_amdgcn_wave_barrier();
            uint32_t old = 0;
            if (valid) old = atomicAdd(&cnt[w][dg >> 1], 1u << (16 * (dg & 1)));
            const uint32_t got = (old >> (16 * (dg & 1))) & 0xffffu;
            if (valid) { mychk++; if (got != before + rk) mybad++; }
        }
    }
    atomicAdd(bad, mybad);
    atomicAdd(checked, mychk);
}

int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    for (int bits = 4; bits <= 10; bits += 3) {
        hipMemset(d, 0, 16);
        order_kernel<<<2048, 256>>>(bits, 2000, 12345 + bits, d, d + 1);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("bits %d: %llu of %llu ranks out of lane order\n", bits, h[0], h[1]);
    }
    return 0;
}
