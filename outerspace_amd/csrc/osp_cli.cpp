// osp_spgemm -- command-line front end with the reference simulator's call shape:
//     ./simulator A.mtx B.mtx          (SimSpGEMM.cpp:819-825; operates on A * B^T, :852-856)
// Prints the same header lines the reference prints (sizes :864-867, "mul flops ref" :891, the
// " -- <caption>: <s> s" timers :30) and, instead of simulated accelerator cycles (:893-894, out of
// scope), the measured GPU product.  Extra flags: --f64, --no-transpose-b, --out C.mtx, --device N, --gpus N (the
// k-sharded product over N GPUs of this node, osp_spgemm_csc_csr_multi; with fewer than N GPUs visible the ranks share them).
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/outerspace_spgemm.h"

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// --gpus N: read and convert on the host (readcoo + coo2csr<true> / coo2csr, SimSpGEMM.cpp:844-880), then ONE call multiplies
// over N ranks.  Ranks are spread over the visible GPUs; with fewer GPUs than ranks they share them.
template <class T>
static int multi_gpu_t(const char *const paths[2], osp_dtype_t dtype, int transpose_b, int gpus, const char *out) {
    uint64_t nr[2], nc[2], nz[2];
    uint32_t *rows[2] = {nullptr, nullptr}, *cols[2] = {nullptr, nullptr};
    double *vals[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) {
        const int st = osp_mtx_read(paths[i], 0, &nr[i], &nc[i], &nz[i], &rows[i], &cols[i], &vals[i]);
        if (st) { fprintf(stderr, "error %d: %s\n", st, osp_last_error_string()); return 1; }
    }
    if (transpose_b) { std::swap(nr[1], nc[1]); std::swap(rows[1], cols[1]); }
    if (nc[0] != nr[1]) { fprintf(stderr, "error %d: inner dimensions differ\n", OSP_ERR_DIM); return 1; }
    const uint64_t M = nr[0], K = nc[0], N = nc[1];
    std::vector<T> av(nz[0] ? nz[0] : 1), bv(nz[1] ? nz[1] : 1), acv(av.size()), bcv(bv.size());
    for (uint64_t i = 0; i < nz[0]; i++) av[i] = (T)vals[0][i];
    for (uint64_t i = 0; i < nz[1]; i++) bv[i] = (T)vals[1][i];
    std::vector<int64_t> ap(K + 1), bp(K + 1);
    std::vector<uint32_t> ai(av.size()), bi(bv.size());
    double t0 = now();
    int st = sizeof(T) == 4 ? osp_coo_to_compressed_f32(1, K, nz[0], rows[0], cols[0], (const float *)av.data(), ap.data(), ai.data(), (float *)acv.data())
                            : osp_coo_to_compressed_f64(1, K, nz[0], rows[0], cols[0], (const double *)av.data(), ap.data(), ai.data(), (double *)acv.data());
    if (!st)
        st = sizeof(T) == 4 ? osp_coo_to_compressed_f32(0, K, nz[1], rows[1], cols[1], (const float *)bv.data(), bp.data(), bi.data(), (float *)bcv.data())
                            : osp_coo_to_compressed_f64(0, K, nz[1], rows[1], cols[1], (const double *)bv.data(), bp.data(), bi.data(), (double *)bcv.data());
    for (int i = 0; i < 2; i++) { osp_host_free(rows[i]); osp_host_free(cols[i]); osp_host_free(vals[i]); }
    if (st) {
        fprintf(stderr, "error %d: %s\n", st, osp_last_error_string());
        return st == OSP_ERR_DUPLICATE ? 233 : 1;  // reference: uncaught throw(233)
    }
    printf(" -- COO2CSR: %g s\n", now() - t0);
    // ranks over the visible GPUs: probe ordinals with single-rank contexts until one fails
    int ndev = 0;
    for (; ndev < gpus; ndev++) {
        osp_context_t probe;
        if (osp_context_create(ndev, &probe) != OSP_OK) break;
        osp_context_destroy(probe);
    }
    if (ndev == 0) { fprintf(stderr, "error: %s\n", osp_last_error_string()); return 1; }
    if (ndev < gpus) fprintf(stderr, "note: %d GPU(s) visible for %d ranks: ranks share GPUs\n", ndev, gpus);
    std::vector<int> devices(gpus);
    for (int g = 0; g < gpus; g++) devices[g] = g % ndev;
    osp_multi_context_t mc = nullptr;
    osp_multi_result_t res = nullptr;
    t0 = now();
    st = osp_spgemm_csc_csr_multi(devices.data(), gpus, dtype, M, K, N, ap.data(), ai.data(), acv.data(), bp.data(), bi.data(), bcv.data(), nullptr,
                                  &mc, &res);
    if (st) { fprintf(stderr, "error %d: %s\n", st, osp_last_error_string()); return st == OSP_ERR_DUPLICATE ? 233 : 1; }
    const double wall = now() - t0;
    osp_multi_info_t info;
    osp_multi_result_info(res, &info);
    printf("mul flops ref = %" PRIu64 "\n", info.partials);
    printf(" -- SpGEMM (%d ranks: upload + product): %g s\n", gpus, wall);
    printf("GPU x%d: nnz(C) = %" PRIu64 ", product %.3f ms (slabs uploaded in %.3f ms), %.3f MB exchanged, %.3f G partials/s, %.3f M nnz/s\n", gpus,
           info.nnz_c, info.ms_total, info.ms_upload, info.bytes_exchanged / 1e6, info.ms_total > 0 ? info.partials / info.ms_total * 1e-6 : 0.0,
           info.ms_total > 0 ? info.nnz_c / info.ms_total * 1e-3 : 0.0);
    for (int g = 0; g < info.nranks; g++) {
        const osp_multi_rank_info_t &r = info.rank[g];
        printf("  rank %d (device %d): k [%" PRIu64 ",%" PRIu64 ") rows [%" PRIu64 ",%" PRIu64 ") %" PRIu64 " partials formed, %" PRIu64
               " merged, %.3f MB sent, %.3f ms (symbolic %.3f, multiply kernels %.3f, merge %.3f)\n",
               g, r.device, r.k_begin, r.k_end, r.row_begin, r.row_end, r.partials_local, r.records_received, r.bytes_sent / 1e6, r.ms_total,
               r.ms_symbolic, r.ms_multiply_kernel, r.ms_merge);
    }
    int rc = 0;
    if (out) {
        std::vector<int64_t> rp(M + 1);
        std::vector<uint32_t> ci(info.nnz_c ? info.nnz_c : 1);
        std::vector<T> cv(ci.size());
        st = osp_multi_result_copy_csr(res, rp.data(), ci.data(), cv.data());
        FILE *f = st ? nullptr : fopen(out, "w");
        if (!f) { fprintf(stderr, "error: cannot write %s\n", out); rc = 1; }
        else {
            fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%%\n%" PRIu64 " %" PRIu64 " %" PRIu64 "\n", M, N, info.nnz_c);
            for (uint64_t row = 0; row < M; row++)
                for (int64_t i = rp[row]; i < rp[row + 1]; i++)
                    fprintf(f, sizeof(T) == 4 ? "%" PRIu64 " %u %.9g\n" : "%" PRIu64 " %u %.17g\n", row + 1, ci[i] + 1, (double)cv[i]);
            fclose(f);
        }
    }
    osp_multi_result_destroy(res);
    osp_multi_context_destroy(mc);
    return rc;
}
static int multi_gpu(const char *const paths[2], osp_dtype_t dtype, int transpose_b, int gpus, const char *out) {
    if (gpus > OSP_MULTI_MAX_RANKS) { fprintf(stderr, "error: at most %d ranks\n", OSP_MULTI_MAX_RANKS); return 2; }
    return dtype == OSP_F32 ? multi_gpu_t<float>(paths, dtype, transpose_b, gpus, out) : multi_gpu_t<double>(paths, dtype, transpose_b, gpus, out);
}

int main(int argc, char *argv[]) {
    const char *paths[2] = {nullptr, nullptr};
    const char *out = nullptr;
    osp_dtype_t dtype = OSP_F32;  // reference value_t is float (common.h:8)
    int transpose_b = 1, device = 0, npos = 0, gpus = 1;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--f64") dtype = OSP_F64;
        else if (a == "--f32") dtype = OSP_F32;
        else if (a == "--no-transpose-b") transpose_b = 0;
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (npos < 2) paths[npos++] = argv[i];
    }
    if (npos < 2) {  // the reference dereferences argv[1], argv[2] unchecked (:824-825)
        fprintf(stderr, "usage: %s A.mtx B.mtx [--f64] [--no-transpose-b] [--out C.mtx] [--device N] [--gpus N]\n", argv[0]);
        return 2;
    }
    // sizes, as the reference prints them (labels swapped there too, :866)
    double t0 = now();
    for (int i = 0; i < 2; i++) {
        uint64_t nr, nc, nz;
        uint32_t *r, *c;
        double *v;
        int st = osp_mtx_read(paths[i], 0, &nr, &nc, &nz, &r, &c, &v);
        if (st) { fprintf(stderr, "error %d: %s\n", st, osp_last_error_string()); return 1; }
        if (i == 1 && transpose_b) { uint64_t t = nr; nr = nc; nc = t; }
        printf("NCol = %" PRIu64 ", NRow = %" PRIu64 ", NNZ = %" PRIu64 "\n", nr, nc, nz);
        osp_host_free(r); osp_host_free(c); osp_host_free(v);
    }
    printf(" -- Read Matrix: %g s\n", now() - t0);

    if (gpus > 1) return multi_gpu(paths, dtype, transpose_b, gpus, out);
    osp_context_t ctx;
    int st = osp_context_create(device, &ctx);
    if (st) { fprintf(stderr, "error %d: %s\n", st, osp_last_error_string()); return 1; }
    osp_result_t res;
    t0 = now();
    st = osp_spgemm_mtx(ctx, dtype, paths[0], paths[1], transpose_b, nullptr, &res);
    if (st) {
        fprintf(stderr, "error %d: %s\n", st, osp_last_error_string());
        osp_context_destroy(ctx);
        return st == OSP_ERR_DUPLICATE ? 233 : 1;  // reference: uncaught throw(233)
    }
    double wall = now() - t0;
    osp_result_info_t info;
    osp_result_info(res, &info);
    printf("mul flops ref = %" PRIu64 "\n", info.partials);
    printf(" -- SpGEMM (read+convert+GPU): %g s\n", wall);
    printf("GPU: nnz(C) = %" PRIu64 ", %.3f ms (symbolic %.3f, multiply %.3f, merge %.3f, compact %.3f), "
           "%.3f G partials/s, %.3f M nnz/s\n",
           info.nnz_c, info.ms_total, info.ms_symbolic, info.ms_multiply, info.ms_merge, info.ms_compact,
           info.ms_total > 0 ? info.partials / info.ms_total * 1e-6 : 0.0,
           info.ms_total > 0 ? info.nnz_c / info.ms_total * 1e-3 : 0.0);
    if (out) {
        st = osp_result_write_mtx(res, out);
        if (st) fprintf(stderr, "error %d: %s\n", st, osp_last_error_string());
    }
    osp_result_destroy(res);
    osp_context_destroy(ctx);
    return st ? 1 : 0;
}
