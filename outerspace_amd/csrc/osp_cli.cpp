// osp_spgemm -- command-line front end with the reference simulator's call shape:
//     ./simulator A.mtx B.mtx          (SimSpGEMM.cpp:819-825; operates on A * B^T, :852-856)
// Prints the same header lines the reference prints (sizes :864-867, "mul flops ref" :891, the
// " -- <caption>: <s> s" timers :30) and, instead of simulated accelerator cycles (:893-894, out of
// scope), the measured GPU product.  Extra flags: --f64, --no-transpose-b, --out C.mtx, --device N.
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/outerspace_spgemm.h"

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char *argv[]) {
    const char *paths[2] = {nullptr, nullptr};
    const char *out = nullptr;
    osp_dtype_t dtype = OSP_F32;  // reference value_t is float (common.h:8)
    int transpose_b = 1, device = 0, npos = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--f64") dtype = OSP_F64;
        else if (a == "--f32") dtype = OSP_F32;
        else if (a == "--no-transpose-b") transpose_b = 0;
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (npos < 2) paths[npos++] = argv[i];
    }
    if (npos < 2) {  // the reference dereferences argv[1], argv[2] unchecked (:824-825)
        fprintf(stderr, "usage: %s A.mtx B.mtx [--f64] [--no-transpose-b] [--out C.mtx] [--device N]\n", argv[0]);
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
