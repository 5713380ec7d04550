// tools/bench_merge.hip -- A/B timing of merge_tiles_kernel variants in ONE process (interleaved rounds).
// Synthetic staging buffer: rows of `rowlen` partial products with random columns in [0, 2^22).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I outerspace_amd/csrc -I tools tools/bench_merge.hip -o tools/bench_merge
#define OSP_MERGE_PROF 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "osp_kernels.h"
#include "osp_merge_runs.h"
using namespace osp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

Part<double> *g_stage;
__global__ void pack_kernel(const uint32_t *pcol, const double *pval, uint64_t n, Part<double> *stage) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stage[i] = Part<double>{pcol[i], pval[i]};
}
__global__ void fill_kernel(uint32_t *pcol, double *pval, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    pcol[i] = (uint32_t)(x & ((1u << 22) - 1));
    pval[i] = 1.0;
}
// chunks of `clen` entries, columns ascending inside a chunk, pseudo-random across chunks
__global__ void fill_sorted_kernel(uint32_t *pcol, uint64_t n, uint32_t clen) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t c = i / clen, j = i % clen;
    uint64_t x = c * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint32_t stride = (1u << 22) / clen;
    pcol[i] = (uint32_t)(x % stride) + (uint32_t)j * stride;
}
// duplicates (DUPWIN=w): every row draws its columns from a window of w columns of its own -- rowlen draws from w values
__global__ void fill_dup_kernel(uint32_t *pcol, uint64_t n, uint32_t rowlen, uint32_t win) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    uint64_t r = (i / rowlen) * 0xD6E8FEB86659FD93ull; r ^= r >> 32;
    pcol[i] = (uint32_t)((r + x % win) & ((1u << 22) - 1));
}
// level-1 look-alike (L1BITS=b): the rows of a tile are consecutive column ranges of width 2^b / rpt, key = col - 0
__global__ void fill_l1_kernel(uint32_t *pcol, uint64_t n, uint32_t rowlen, uint32_t rpt, uint32_t width) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint32_t r = (uint32_t)((i / rowlen) % rpt);
    pcol[i] = r * width + (uint32_t)(x % width);
}
__global__ void desc_l1_kernel(TileDesc *desc, uint32_t ntiles, uint32_t kbits) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntiles) { desc[i].kbits = kbits; desc[i].cbase = 0; }
}
__global__ void chunks_kernel(uint64_t *chunk_start, uint64_t nchunks, uint32_t clen) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nchunks) chunk_start[i] = i * clen;
}
__global__ void arow_kernel(uint32_t *arow, uint64_t M, uint32_t cpr) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= M) arow[i] = (uint32_t)(i * cpr);
}
uint32_t *g_arow; uint64_t *g_chunk_start;
template <int NT, int ABL>
float run_runs(uint32_t *tile_rows, uint32_t ntiles, uint64_t M, uint64_t *row_off, uint32_t *pcol, double *pval,
          uint32_t *heavy, uint64_t *status, uint32_t *ticket, uint64_t *outn, int64_t *rowptr, uint32_t *ccol, double *cval) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemsetAsync(status, 0, (uint64_t)ntiles * 8, 0));
    CK(hipMemsetAsync(ticket, 0, sizeof(uint32_t), 0));
    CK(hipMemsetAsync(outn, 0, 16, 0));
    CK(hipEventRecord(a, 0));
    merge_runs_kernel<double, NT, ABL><<<ntiles, NT, 0, 0>>>(tile_rows, ntiles, M, row_off, 0, g_arow, g_chunk_start, g_stage, heavy,
                                                             status, ticket, outn, rowptr, ccol, cval, outn + 1);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms;
}
__global__ void rows_kernel(uint64_t *row_off, uint64_t M, uint32_t rowlen) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= M) row_off[i] = i * rowlen;
}
__global__ void tiles_kernel(uint32_t *tile_rows, uint32_t ntiles, uint32_t rpt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntiles) tile_rows[i] = i * rpt;
}

TileDesc *g_desc; uint32_t g_grid = 512;
uint32_t g_shards = 1;
template <int NT, int ABL, int CAP, int MAXWG, int SHARDS>
float run_sharded(uint32_t *tile_rows, uint32_t ntiles, uint64_t M, uint64_t *row_off, uint32_t *pcol, double *pval,
          uint32_t *heavy, uint64_t *status, uint32_t *ticket, uint64_t *outn, int64_t *rowptr, uint32_t *ccol, double *cval);
template <int NT, int ABL, int CAP = 2 * NT * 3, int MAXWG = 5>
float run(uint32_t *tile_rows, uint32_t ntiles, uint64_t M, uint64_t *row_off, uint32_t *pcol, double *pval,
          uint32_t *heavy, uint64_t *status, uint32_t *ticket, uint64_t *outn, int64_t *rowptr, uint32_t *ccol, double *cval) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemsetAsync(status, 0, (uint64_t)ntiles * 8, 0));
    CK(hipMemsetAsync(ticket, 0, 64 * kTicketStride * sizeof(uint32_t), 0));
    CK(hipMemsetAsync(outn, 0, 16, 0));
    CK(hipEventRecord(a, 0));
    MergeLevels<double> lv{};
    lv.stage[0] = g_stage; lv.row_off[0] = row_off; lv.base[0] = 0; lv.c_rowptr[0] = rowptr; lv.heavy_nnz[0] = heavy;
    const uint32_t grid = g_grid * (uint32_t)merge_wgs_per_cu<double, NT, CAP, MAXWG>() / 2;  // g_grid assumes 2 per CU
    merge_tiles_kernel<double, NT, ABL, CAP, MAXWG><<<ntiles < grid ? ntiles : grid, NT, 0, 0>>>(g_desc, ntiles, lv, 22, status, ticket, outn,
                                                                                         ccol, cval, outn + 1, ChunkTable<double>{}, g_shards,
                                                                                         ticket + 40 * kTicketStride);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms;
}

template <int NT, int ABL, int CAP, int MAXWG, int SHARDS>
float run_sharded(uint32_t *tile_rows, uint32_t ntiles, uint64_t M, uint64_t *row_off, uint32_t *pcol, double *pval,
          uint32_t *heavy, uint64_t *status, uint32_t *ticket, uint64_t *outn, int64_t *rowptr, uint32_t *ccol, double *cval) {
    g_shards = SHARDS;
    const float ms = run<NT, ABL, CAP, MAXWG>(tile_rows, ntiles, M, row_off, pcol, pval, heavy, status, ticket, outn, rowptr, ccol, cval);
    g_shards = 1;
    return ms;
}

__global__ void checksum_kernel(const int64_t *rowptr, uint64_t M, const uint32_t *ccol, const double *cval, uint64_t nnz,
                                unsigned long long *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = 0;
    if (i <= M) h += (unsigned long long)rowptr[i] * 0x9E3779B97F4A7C15ull + i;
    if (i < nnz) h += ((unsigned long long)ccol[i] + 1) * (0xBF58476D1CE4E5B9ull ^ i) + (unsigned long long)__double_as_longlong(cval[i]);
    for (int d = 32; d > 0; d >>= 1) h += __shfl_down(h, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

int main(int argc, char **argv) {
    const uint32_t rowlen = argc > 1 ? atoi(argv[1]) : 256;
    const uint32_t rpt = argc > 2 ? atoi(argv[2]) : 9;          // rows per tile
    const uint64_t M = argc > 3 ? atoll(argv[3]) : (1u << 20);
    const uint64_t P = M * rowlen;
    const uint32_t ntiles = (uint32_t)((M + rpt - 1) / rpt);
    uint32_t *pcol, *tile_rows, *heavy, *ticket, *ccol; double *pval, *cval; uint64_t *row_off, *status, *outn; int64_t *rowptr;
    CK(hipMalloc(&pcol, P * 4)); CK(hipMalloc(&pval, P * 8)); CK(hipMalloc(&ccol, (P + 4096ull * 0) * 4 + (uint64_t)ntiles * 3072 * 4));
    CK(hipMalloc(&cval, (uint64_t)ntiles * 3072 * 8 + P * 8)); CK(hipMalloc(&row_off, (M + 1) * 8)); CK(hipMalloc(&tile_rows, ntiles * 4));
    CK(hipMalloc(&heavy, M * 4)); CK(hipMalloc(&status, (uint64_t)ntiles * 8)); CK(hipMalloc(&ticket, 64 * kTicketStride * sizeof(uint32_t))); CK(hipMalloc(&outn, 16));
    CK(hipMalloc(&rowptr, (M + 1) * 8));
    const uint32_t clen = argc > 4 ? atoi(argv[4]) : 16;
    fill_kernel<<<(unsigned)((P + 255) / 256), 256>>>(pcol, pval, P);
    fill_sorted_kernel<<<(unsigned)((P + 255) / 256), 256>>>(pcol, P, clen);
    if (getenv("DUPWIN")) fill_dup_kernel<<<(unsigned)((P + 255) / 256), 256>>>(pcol, P, rowlen, (uint32_t)atoi(getenv("DUPWIN")));
    const uint32_t l1bits = getenv("L1BITS") ? atoi(getenv("L1BITS")) : 0;
    if (l1bits) fill_l1_kernel<<<(unsigned)((P + 255) / 256), 256>>>(pcol, P, rowlen, rpt, (1u << l1bits) / rpt);
    CK(hipMalloc(&g_arow, (M + 1) * 4)); CK(hipMalloc(&g_chunk_start, (P / clen + 2) * 8));
    chunks_kernel<<<(unsigned)((P / clen + 256) / 256), 256>>>(g_chunk_start, P / clen, clen);
    arow_kernel<<<(unsigned)((M + 256) / 256), 256>>>(g_arow, M, rowlen / clen);
    rows_kernel<<<(unsigned)((M + 256) / 256), 256>>>(row_off, M, rowlen);
    tiles_kernel<<<(ntiles + 255) / 256, 256>>>(tile_rows, ntiles, rpt);
    CK(hipMalloc(&g_stage, P * sizeof(Part<double>)));
    pack_kernel<<<(unsigned)((P + 255) / 256), 256>>>(pcol, pval, P, g_stage);
    CK(hipMalloc(&g_desc, (uint64_t)ntiles * sizeof(TileDesc)));
    tile_desc_kernel<1 << 20><<<(ntiles + 255) / 256, 256>>>(tile_rows, ntiles, M, row_off, 0, 0u, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, g_desc);
    if (l1bits) desc_l1_kernel<<<(ntiles + 255) / 256, 256>>>(g_desc, ntiles, l1bits);
    if (getenv("GRID")) g_grid = atoi(getenv("GRID"));
    CK(hipDeviceSynchronize());
    printf("P=%llu partials, %u tiles of %u rows x %u (%u per tile), algorithmic bytes %.2f GB\n", (unsigned long long)P, ntiles, rpt,
           rowlen, rpt * rowlen, (12.0 * P * 2) / 1e9);
#define ARGS tile_rows, ntiles, M, row_off, pcol, pval, heavy, status, ticket, outn, rowptr, ccol, cval
    struct V { const char *name; uint32_t cap; float (*fn)(uint32_t *, uint32_t, uint64_t, uint64_t *, uint32_t *, double *, uint32_t *, uint64_t *, uint32_t *, uint64_t *, int64_t *, uint32_t *, double *); };
    std::vector<V> vs = {
        {"radix NT256 cap1536 full", 1536, run<256, 0, 1536>}, {"radix NT256 cap1536 nosort", 1536, run<256, 1, 1536>},
        {"radix NT256 cap1536 nolb", 1536, run<256, 2, 1536>}, {"radix NT256 cap1536 nosort+nolb", 1536, run<256, 3, 1536>},
        {"radix NT256 cap1536 full 8 shards", 1536, run_sharded<256, 0, 1536, 5, 8>}, {"radix NT256 cap1536 full 4 shards", 1536, run_sharded<256, 0, 1536, 5, 4>},
        {"radix NT256 cap1536 full 16 shards", 1536, run_sharded<256, 0, 1536, 5, 16>}, {"radix NT256 cap1536 full 2 shards", 1536, run_sharded<256, 0, 1536, 5, 2>},
        {"radix NT256 cap1536 nolb 8 shards", 1536, run_sharded<256, 2, 1536, 5, 8>},
        {"radix NT256 cap1536 latecount 8 shards", 1536, run_sharded<256, 8, 1536, 5, 8>},
        {"radix NT256 cap1536 latecount", 1536, run<256, 8, 1536>},
        {"radix NT256 cap1536 static tiles (no ticket)", 1536, run<256, 4, 1536>},
        {"radix NT256 cap1536 static nolb", 1536, run<256, 6, 1536>}, {"radix NT256 cap1536 static nosort+nolb", 1536, run<256, 7, 1536>},
        {"radix NT256 cap1536 b16 counters", 1536, run<256, 128, 1536>}, {"radix NT256 cap1536 b32 hash init", 1536, run<256, 256, 1536>},
        {"radix NT256 cap1536 old lds ops", 1536, run<256, 128 + 256, 1536>},
        {"radix NT256 cap1536 4wg", 1536, run<256, 0, 1536, 4>}, {"radix NT256 cap1264 6wg", 1264, run<256, 0, 1264, 6>}, {"radix NT256 cap1792 full", 1792, run<256, 0, 1792>},
        {"radix NT256 cap2048 full", 2048, run<256, 0, 2048>}, {"radix NT256 cap1280 6wg", 1280, run<256, 0, 1280, 6>},
        {"radix NT512 cap3072 full", 3072, run<512, 0, 3072>},
        {"radix NT384 cap2304 full", 2304, run<384, 0, 2304>}, {"radix NT320 cap1920 full", 1920, run<320, 0, 1920>},
        {"runs NT1024 full", 3072, run_runs<1024, 0>},
    };
    if (getenv("CHECK_GRIDS")) {
        // the output must not depend on how many workgroups run or in which order they take tickets
        unsigned long long *d_sum; CK(hipMalloc(&d_sum, 8));
        unsigned long long ref = 0; uint64_t ref_total = 0;
        for (uint32_t g : {512u, 1u, 7u, 64u, 511u, 513u, 1000u, 4096u, 100000u}) {
            g_grid = g;
            g_shards = getenv("SHARDS") ? atoi(getenv("SHARDS")) : 1;
            float ms = run<256, 0, 1536>(ARGS);
            g_shards = 1;
            uint64_t h_out[2]; CK(hipMemcpy(h_out, outn, 16, hipMemcpyDeviceToHost));
            CK(hipMemset(d_sum, 0, 8));
            const uint64_t n = std::max<uint64_t>(M + 1, h_out[1]);
            checksum_kernel<<<(unsigned)((n + 255) / 256), 256>>>(rowptr, M, ccol, cval, h_out[1], d_sum);
            unsigned long long hs; CK(hipMemcpy(&hs, d_sum, 8, hipMemcpyDeviceToHost));
            if (g == 512u) { ref = hs; ref_total = h_out[1]; }
            printf("grid %6u: %8.3f ms  nnz %llu  checksum %016llx  %s\n", g, ms, (unsigned long long)h_out[1], hs,
                   (hs == ref && h_out[1] == ref_total) ? "same" : "DIFFERENT");
        }
        g_grid = 512;
    }
    if (getenv("PROF")) {
        // where a workgroup's time goes: cycles between marks, summed over workgroups (thread 0's clock)
        static const char *names[11] = {"wait for tile data", "stage + hash count", "publish", "pass: rank", "pass: digit scan",
                                        "pass: scatter", "head flags + scan", "look-back + ticket", "run sums", "compact", "next request + output"};
        unsigned long long z[16] = {0}, h[16];
        for (int ab = 0; ab < 2; ab++) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(osp_merge_prof), z, sizeof(z)));
            float ms = ab == 0 ? run<256, 0, 1536>(ARGS) : run<256, 2, 1536>(ARGS);
            CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(osp_merge_prof), sizeof(h)));
            unsigned long long tot = 0;
            for (int k = 0; k < 11; k++) tot += h[k];
            printf("%s: %.3f ms, %u tiles; cycles per tile per workgroup (share):\n", ab == 0 ? "full" : "no look-back", ms, ntiles);
            for (int k = 0; k < 11; k++) printf("  %-24s %9.0f  %5.1f%%\n", names[k], (double)h[k] / ntiles, 100.0 * h[k] / tot);
            printf("  %-24s %9.0f\n", "total", (double)tot / ntiles);
        }
    }
    {   // variants whose tile capacity is below this run's tile size do not apply
        std::vector<V> keep;
        for (auto &v : vs) if (v.cap >= rpt * rowlen) keep.push_back(v);
        vs = keep;
    }
    if (getenv("ONLY")) {  // run one variant alone (substring match), announcing it first
        std::vector<V> keep;
        for (auto &v : vs) if (strstr(v.name, getenv("ONLY"))) keep.push_back(v);
        vs = keep;
    }
    std::vector<std::vector<float>> t(vs.size());
    for (int round = 0; round < 5; round++)
        for (size_t v = 0; v < vs.size(); v++) {
            if (getenv("ONLY")) { printf("round %d: %s\n", round, vs[v].name); fflush(stdout); }
            t[v].push_back(vs[v].fn(ARGS));
        }
    for (size_t v = 0; v < vs.size(); v++) {
        std::sort(t[v].begin(), t[v].end());
        printf("%-28s median %8.3f ms  min %8.3f ms   %7.1f GB/s (alg)\n", vs[v].name, t[v][2], t[v][0], 24.0 * P / (t[v][2] * 1e-3) / 1e9);
    }
    return 0;
}
