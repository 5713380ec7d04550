// osp_host.cpp -- host side of the reference CLI's data flow (no GPU code):
//   osp_mtx_read               readcoo                    SimSpGEMM.cpp:55-100
//   osp_coo_to_compressed_*    coo2csr<transpose>+dupcheck SimSpGEMM.cpp:102-152, 43-53
//   osp_spgemm_mtx             main()                      SimSpGEMM.cpp:819-891
//   osp_result_write_mtx       the format util.py:61-62 (scipy.io.mmwrite) produces
#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/outerspace_spgemm.h"
#include "osp_internal.h"

#include <cstdarg>
#include <string>

namespace osp {
// last error of the calling thread (osp_last_error_string); shared with osp_api.hip through osp_internal.h
thread_local std::string g_last_error;
int fail(int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}
}  // namespace osp

using osp::fail;

extern "C" {

void osp_host_free(void *p) { free(p); }
const char *osp_last_error_string(void) { return osp::g_last_error.c_str(); }

// One line of the file, NUL-terminated in `buf` (the rules below are written for C strings; a line never reaches into
// the next one).  Returns false for a line the reference skips (:66-77): first non-blank char is '%', or nothing but blanks.
static inline const char *kept_line(const char *b, const char *e, std::string &buf) {
    buf.assign(b, e);
    const char *p = buf.c_str();
    while (*p == ' ' || *p == '\t') p++;
    if (*p == '%' || *p == '\0' || *p == '\n' || *p == '\r') return nullptr;
    return p;
}
struct MtxPiece {   // what one thread parsed: entries of its byte range, in file order
    std::vector<uint32_t> r, c;
    std::vector<double> v;
    bool bad_alloc = false;
};
// entries "row col [val]" of the lines that START in [b, e) (the last one may end beyond e, never beyond `end`)
static void parse_entries(const char *b, const char *e, const char *end, int symmetric, MtxPiece &out) {
    try {
        std::string buf;
        const char *p = b;
        while (p < e) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end;
            const char *q = kept_line(p, le, buf);
            p = nl ? nl + 1 : end;
            if (!q) continue;
            // strtoull/strtod instead of sscanf: same accepted syntax, much faster
            char *stop = nullptr;
            unsigned long long row = strtoull(q, &stop, 10);
            unsigned long long col = 0;
            double val = 1.0;  // pattern entry (:92-93)
            if (stop != q) {
                const char *t = stop;
                col = strtoull(t, &stop, 10);
                if (stop != t) {
                    t = stop;
                    double x = strtod(t, &stop);
                    if (stop != t) val = x;
                }
            }
            out.r.push_back((uint32_t)(row - 1));  // 1-based -> 0-based (:94)
            out.c.push_back((uint32_t)(col - 1));
            out.v.push_back(val);
            if (symmetric && row != col) {  // :95-96
                out.r.push_back((uint32_t)(col - 1));
                out.c.push_back((uint32_t)(row - 1));
                out.v.push_back(val);
            }
        }
    } catch (const std::bad_alloc &) { out.bad_alloc = true; }
}

// The reference reads its files with one thread, getline + sscanf per entry (TIMER("Read Matrix"), :844-850: minutes for
// the 65 M lines of a scale-22 operand).  Here the file is mapped and the lines behind the header are parsed by several
// threads, each on a byte range that starts at a line start; the pieces are concatenated in file order, so the result is
// what the single-threaded loop gives.  OSP_PARSE_THREADS overrides the thread count (default: one per 4 MB, at most 32
// and at most the hardware's).
int osp_mtx_read(const char *path, int symmetric, uint64_t *nrow, uint64_t *ncol, uint64_t *nnz,
                 uint32_t **rows, uint32_t **cols, double **vals) {
    if (!path || !nrow || !ncol || !nnz || !rows || !cols || !vals) return fail(OSP_ERR_ARG, "null argument");
    *rows = nullptr; *cols = nullptr; *vals = nullptr;
    *nrow = *ncol = *nnz = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(OSP_ERR_IO, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); return fail(OSP_ERR_IO, "cannot read %s", path); }
    const size_t bytes = (size_t)sb.st_size;
    const char *data = nullptr;
    if (bytes) {
        void *m = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); return fail(OSP_ERR_IO, "cannot map %s", path); }
        data = (const char *)m;
        (void)madvise(m, bytes, MADV_SEQUENTIAL);
    }
    close(fd);
    int st = OSP_OK;
    try {  // nothing thrown here may cross the C boundary
        const char *end = data + bytes;
        // the first kept line is "rows cols nnz" (:79-88)
        unsigned long long NR = 0, NC = 0, NZ = 0;
        const char *p = data;
        {
            std::string buf;
            while (p < end) {
                const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                const char *q = kept_line(p, nl ? nl : end, buf);
                p = nl ? nl + 1 : end;
                if (q) { sscanf(q, "%llu %llu %llu", &NR, &NC, &NZ); break; }
            }
        }
        const size_t body = (size_t)(end - p);
        unsigned nthreads = 1;
        if (const char *env = getenv("OSP_PARSE_THREADS")) nthreads = (unsigned)std::max(1, atoi(env));
        else nthreads = (unsigned)std::min<size_t>(std::min<size_t>(32, std::max(1u, std::thread::hardware_concurrency())), body / (4u << 20) + 1);
        std::vector<MtxPiece> pieces(nthreads);
        // an entry takes at least four bytes of the file ("1 1\n"): a header that announces more than the file can hold
        // (or garbage) must not drive the reservation
        const unsigned long long want = std::min<unsigned long long>(NZ, bytes / 4 + 1) * (symmetric ? 2ull : 1ull);
        // byte ranges that start at line starts
        std::vector<const char *> cut(nthreads + 1, end);
        cut[0] = p;
        for (unsigned t = 1; t < nthreads; t++) {
            const char *c = p + body / nthreads * t;
            if (c <= cut[t - 1]) { cut[t] = cut[t - 1]; continue; }
            const char *nl = (const char *)memchr(c - 1, '\n', (size_t)(end - (c - 1)));   // the line break at or behind c - 1
            cut[t] = nl ? nl + 1 : end;
        }
        for (unsigned t = 0; t < nthreads; t++) {
            const size_t r = want / nthreads + 16;
            pieces[t].r.reserve(r); pieces[t].c.reserve(r); pieces[t].v.reserve(r);
        }
        if (nthreads == 1) {
            parse_entries(cut[0], cut[1], end, symmetric, pieces[0]);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthreads; t++)
                th.emplace_back(parse_entries, cut[t], cut[t + 1], end, symmetric, std::ref(pieces[t]));
            for (auto &x : th) x.join();
        }
        size_t n = 0;
        for (auto &pc : pieces) { if (pc.bad_alloc) throw std::bad_alloc(); n += pc.r.size(); }
        *rows = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
        *cols = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
        *vals = (double *)malloc((n ? n : 1) * sizeof(double));
        if (!*rows || !*cols || !*vals) {
            st = fail(OSP_ERR_ALLOC, "host allocation failed");
        } else {
            size_t o = 0;
            for (auto &pc : pieces) {
                const size_t m = pc.r.size();
                if (m) {
                    memcpy(*rows + o, pc.r.data(), m * sizeof(uint32_t));
                    memcpy(*cols + o, pc.c.data(), m * sizeof(uint32_t));
                    memcpy(*vals + o, pc.v.data(), m * sizeof(double));
                }
                o += m;
            }
            *nrow = NR; *ncol = NC; *nnz = n;
        }
    } catch (const std::bad_alloc &) {
        st = fail(OSP_ERR_ALLOC, "host allocation failed while reading %s", path);
    } catch (const std::exception &e) {
        st = fail(OSP_ERR_ALLOC, "%s while reading %s", e.what(), path);
    }
    if (data) (void)munmap((void *)data, bytes);
    if (st != OSP_OK) {  // hand nothing half-made to the caller
        free(*rows); free(*cols); free(*vals);
        *rows = nullptr; *cols = nullptr; *vals = nullptr;
    }
    return st;
}

}  // extern "C"

namespace {

// Counting sort by segment, then order each segment by the inner index; adjacent equal inner
// indices are the reference's duplicate condition (dupcheck after the sort, :123).
template <class T>
int coo_to_compressed(int by_col, uint64_t nseg, uint64_t nnz, const uint32_t *rows, const uint32_t *cols,
                      const T *vals, int64_t *ptr, uint32_t *idx, T *out_vals) {
    if (!ptr || (nnz && (!rows || !cols || !vals || !idx || !out_vals))) return fail(OSP_ERR_ARG, "null argument");
    const uint32_t *seg = by_col ? cols : rows;
    const uint32_t *in = by_col ? rows : cols;
    std::fill(ptr, ptr + nseg + 1, (int64_t)0);
    for (uint64_t i = 0; i < nnz; i++) {
        if (seg[i] >= nseg) return fail(OSP_ERR_RANGE, "entry %" PRIu64 ": segment index %u >= %" PRIu64, i, seg[i], nseg);
        ptr[seg[i] + 1]++;
    }
    for (uint64_t s = 0; s < nseg; s++) ptr[s + 1] += ptr[s];
    std::vector<int64_t> cursor(ptr, ptr + nseg);
    for (uint64_t i = 0; i < nnz; i++) {
        int64_t o = cursor[seg[i]]++;
        idx[o] = in[i];
        out_vals[o] = vals[i];
    }
    std::vector<std::pair<uint32_t, T>> tmp;
    for (uint64_t s = 0; s < nseg; s++) {
        const int64_t b = ptr[s], e = ptr[s + 1];
        bool sorted = true;
        for (int64_t i = b + 1; i < e; i++) if (idx[i] <= idx[i - 1]) { sorted = false; break; }
        if (!sorted) {
            tmp.resize(e - b);
            for (int64_t i = b; i < e; i++) tmp[i - b] = {idx[i], out_vals[i]};
            std::stable_sort(tmp.begin(), tmp.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
            for (int64_t i = b; i < e; i++) { idx[i] = tmp[i - b].first; out_vals[i] = tmp[i - b].second; }
            for (int64_t i = b + 1; i < e; i++)
                if (idx[i] == idx[i - 1])
                    return fail(OSP_ERR_DUPLICATE, "duplicate coordinate in segment %" PRIu64 " (reference: throw(233))", s);
        }
    }
    return OSP_OK;
}

#ifndef OSP_HOST_ONLY  // (the sanitizer build of the host-only code leaves out what calls the GPU entry points)
template <class T>
int spgemm_mtx_t(osp_context_t ctx, osp_dtype_t dtype, const char *pa, const char *pb, int transpose_b,
                 const osp_config_t *cfg, osp_result_t *result) {
    uint64_t nr[2], nc[2], nz[2];
    uint32_t *rows[2] = {nullptr, nullptr}, *cols[2] = {nullptr, nullptr};
    double *vals[2] = {nullptr, nullptr};
    const char *fn[2] = {pa, pb};
    int st = OSP_OK;
    for (int i = 0; i < 2 && st == OSP_OK; i++)
        st = osp_mtx_read(fn[i], 0, &nr[i], &nc[i], &nz[i], &rows[i], &cols[i], &vals[i]);
    if (st == OSP_OK) {
        if (transpose_b) {  // "Workaround: Transpose Matrix 2", :852-856
            std::swap(nr[1], nc[1]);
            std::swap(rows[1], cols[1]);
        }
        if (nc[0] != nr[1]) st = fail(OSP_ERR_DIM, "inner dimensions differ: A is %" PRIu64 "x%" PRIu64 ", B is %" PRIu64 "x%" PRIu64,
                                      nr[0], nc[0], nr[1], nc[1]);
    }
    if (st == OSP_OK) {
        const uint64_t K = nc[0];
        std::vector<T> av(nz[0] ? nz[0] : 1), bv(nz[1] ? nz[1] : 1);
        for (uint64_t i = 0; i < nz[0]; i++) av[i] = (T)vals[0][i];  // value_t(val), :94
        for (uint64_t i = 0; i < nz[1]; i++) bv[i] = (T)vals[1][i];
        // csc = coo2csr<true>(A), csr = coo2csr(B') (:878-879) and the product, all on the device
        st = osp_spgemm_coo(ctx, dtype, nr[0], K, nc[1], nz[0], rows[0], cols[0], av.data(), nz[1], rows[1], cols[1], bv.data(),
                            OSP_HOST, cfg, result);
    }
    for (int i = 0; i < 2; i++) { free(rows[i]); free(cols[i]); free(vals[i]); }
    return st;
}

#endif
}  // namespace

extern "C" {

int osp_coo_to_compressed_f32(int by_col, uint64_t nseg, uint64_t nnz, const uint32_t *rows, const uint32_t *cols,
                              const float *vals, int64_t *ptr, uint32_t *idx, float *out_vals) {
    try { return coo_to_compressed<float>(by_col, nseg, nnz, rows, cols, vals, ptr, idx, out_vals); }
    catch (const std::exception &e) { return fail(OSP_ERR_ALLOC, "%s", e.what()); }
}
int osp_coo_to_compressed_f64(int by_col, uint64_t nseg, uint64_t nnz, const uint32_t *rows, const uint32_t *cols,
                              const double *vals, int64_t *ptr, uint32_t *idx, double *out_vals) {
    try { return coo_to_compressed<double>(by_col, nseg, nnz, rows, cols, vals, ptr, idx, out_vals); }
    catch (const std::exception &e) { return fail(OSP_ERR_ALLOC, "%s", e.what()); }
}

#ifndef OSP_HOST_ONLY
int osp_spgemm_mtx(osp_context_t ctx, osp_dtype_t dtype, const char *path_a, const char *path_b, int transpose_b,
                   const osp_config_t *cfg, osp_result_t *result) {
    if (!ctx || !path_a || !path_b || !result) return fail(OSP_ERR_ARG, "null argument");
    try {
        if (dtype == OSP_F32) return spgemm_mtx_t<float>(ctx, dtype, path_a, path_b, transpose_b, cfg, result);
        if (dtype == OSP_F64) return spgemm_mtx_t<double>(ctx, dtype, path_a, path_b, transpose_b, cfg, result);
    } catch (const std::exception &e) { return fail(OSP_ERR_ALLOC, "%s", e.what()); }
    return fail(OSP_ERR_ARG, "dtype must be OSP_F32 or OSP_F64");
}

int osp_result_write_mtx(osp_result_t r, const char *path) {
    if (!r || !path) return fail(OSP_ERR_ARG, "null argument");
    osp_result_info_t info;
    int st = osp_result_info(r, &info);
    if (st) return st;
    try {
        std::vector<int64_t> rp(info.M + 1);
        std::vector<uint32_t> ci(info.nnz_c ? info.nnz_c : 1);
        std::vector<double> vd;
        std::vector<float> vf;
        void *vp;
        if (info.dtype == OSP_F32) { vf.resize(ci.size()); vp = vf.data(); } else { vd.resize(ci.size()); vp = vd.data(); }
        st = osp_result_copy_csr(r, rp.data(), ci.data(), vp, OSP_HOST);
        if (st) return st;
        FILE *f = fopen(path, "w");
        if (!f) return fail(OSP_ERR_IO, "cannot open %s for writing", path);
        fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%%\n%" PRIu64 " %" PRIu64 " %" PRIu64 "\n", info.M, info.N,
                info.nnz_c);
        for (uint64_t row = 0; row < info.M; row++)
            for (int64_t i = rp[row]; i < rp[row + 1]; i++) {
                if (info.dtype == OSP_F32) fprintf(f, "%" PRIu64 " %u %.9g\n", row + 1, ci[i] + 1, (double)vf[i]);
                else fprintf(f, "%" PRIu64 " %u %.17g\n", row + 1, ci[i] + 1, vd[i]);
            }
        fclose(f);
    } catch (const std::exception &e) { return fail(OSP_ERR_ALLOC, "%s", e.what()); }
    return OSP_OK;
}

#endif  // OSP_HOST_ONLY

}  // extern "C"
