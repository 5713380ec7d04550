// osp_epilogue.h -- what a sparse MLP layer does to a product before it feeds the next one:
//     x = relu(fc(x))            NN_models/models.py:17-31   (fc adds its bias to every element of the dense product)
// on the CSR result of osp_spgemm_*, on the device: out[i,j] = relu(C[i,j] + bias[j]) for every j when there is a bias
// (an entry C lacks counts as 0, so a biased row is dense before the ReLU), for the stored entries only when there is
// none; what comes out as zero is dropped, the rest is a CSR with exact row pointers -- the operand of the next product.
// The reference has no kernel for this (its layers are dense torch ops, models.py:10-31; get_mtx_files.py:76-96 dumps
// the re-sparsified activations as .mtx); this replaces a dense round trip through torch and host scipy.
#pragma once
#include "osp_kernels.h"

namespace osp {

// value of row r at column j (its columns ascend), 0 when absent
template <class T>
__device__ __forceinline__ T csr_at(const uint32_t *__restrict__ col, const T *__restrict__ val, int64_t b, int64_t e, uint32_t j) {
    const uint64_t p = lower_bound_dev(col, (uint64_t)b, (uint64_t)e, (uint64_t)j);
    return (p < (uint64_t)e && col[p] == j) ? val[p] : T(0);
}
template <class T>
__device__ __forceinline__ T epilogue_value(T c, T bias, int relu) {
    const T v = c + bias;
    return (relu && !(v > T(0))) ? T(0) : v;   // relu: max(v, 0); NaN -> 0 like torch.clamp would not -- no NaN reaches here from finite inputs
}
// one wave per row; pass 0 counts the survivors (cnt[r]), pass 1 writes them at out_ptr[r]
template <class T, bool WRITE>
__global__ __launch_bounds__(256) void bias_relu_rows_kernel(const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                             const T *__restrict__ val, uint64_t M, uint64_t N, const T *__restrict__ bias,
                                                             int relu, uint32_t *__restrict__ cnt, const int64_t *__restrict__ out_ptr,
                                                             uint32_t *__restrict__ out_col, T *__restrict__ out_val) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= M) return;
    const unsigned lane = lane_id();
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    uint64_t out = WRITE ? (uint64_t)out_ptr[r] : 0ull;
    uint32_t total = 0;
    if (bias) {
        for (uint64_t j0 = 0; j0 < N; j0 += kWave) {
            const uint64_t j = j0 + lane;
            T v = T(0);
            if (j < N) v = epilogue_value(csr_at(col, val, b, e, (uint32_t)j), bias[j], relu);
            const bool keep = j < N && v != T(0);
            const uint64_t m = __ballot(keep);
            if (WRITE && keep) {
                const uint64_t o = out + (uint64_t)__popcll(m & lanemask_lt());
                out_col[o] = (uint32_t)j;
                out_val[o] = v;
            }
            out += (uint64_t)__popcll(m);
            total += (uint32_t)__popcll(m);
        }
    } else {
        for (int64_t p0 = b; p0 < e; p0 += kWave) {
            const int64_t p = p0 + lane;
            T v = T(0);
            uint32_t c = 0;
            if (p < e) { c = col[p]; v = epilogue_value(val[p], T(0), relu); }
            const bool keep = p < e && v != T(0);
            const uint64_t m = __ballot(keep);
            if (WRITE && keep) {
                const uint64_t o = out + (uint64_t)__popcll(m & lanemask_lt());
                out_col[o] = c;
                out_val[o] = v;
            }
            out += (uint64_t)__popcll(m);
            total += (uint32_t)__popcll(m);
        }
    }
    if (!WRITE && lane == 0) cnt[r] = total;
}
// rows of a CSR as a COO row array (the next product takes COO operands: osp_spgemm_coo)
__global__ void csr_expand_rows_kernel(const int64_t *__restrict__ rowptr, uint64_t M, uint32_t *__restrict__ rows) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= M) return;
    for (int64_t p = rowptr[r] + lane_id(); p < rowptr[r + 1]; p += kWave) rows[p] = (uint32_t)r;
}

// ---- what a plain stream reaches (osp_stream_copy_probe) ----
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, uint64_t n) {
    // one 16-byte element per lane, one workgroup per 4 KB: the shape that reaches the highest rate here (tools/bench_copy:
    // 6.2 TB/s; persistent grid-stride loops with 1-8 loads in flight per lane 4.5-5.5, hipMemcpyAsync 4.8).  NT: non-temporal
    // accesses (the form the library's own streams use for what is written once and read much later)
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
    else dst[i] = src[i];
}

}  // namespace osp
