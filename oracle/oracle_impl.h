/*
 * oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain-C CPU restatement of the reference's outer-product SpGEMM path, included
 * twice by oracle_spgemm.c (once per value type).  Before inclusion define
 *   OSP_T     the value type  (float | double)        -- reference: common.h:8
 *   OSP_SFX   the symbol suffix (f32 | f64)
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/simulator/).  Nothing under outerspace_amd/ may call into this.
 */

#define OSP_CAT_(a, b) a##_##b
#define OSP_CAT(a, b) OSP_CAT_(a, b)
#define OSP_FN(name) OSP_CAT(name, OSP_SFX)

/* One partial product / COO entry.  Reference: COOElement, common.h:18-33. */
typedef struct {
    uint32_t row, col;
    OSP_T val;
} OSP_FN(osp_coo);

/* (row,col) lexicographic order.  Reference: COOElement::operator<, common.h:29-32. */
static int OSP_FN(coo_less_rc)(const OSP_FN(osp_coo) * a, const OSP_FN(osp_coo) * b) {
    return a->row == b->row ? a->col < b->col : a->row < b->row;
}
/* (col,row) order used for the CSC build.  Reference: SimSpGEMM.cpp:113-116. */
static int OSP_FN(coo_less_cr)(const OSP_FN(osp_coo) * a, const OSP_FN(osp_coo) * b) {
    return a->col == b->col ? a->row < b->row : a->col < b->col;
}

/* Stable bottom-up merge sort.  The reference calls the UNSTABLE std::sort
 * (SimSpGEMM.cpp:111-121, :521), so the order among equal keys is
 * implementation-defined there; the oracle pins it to "input order", which for
 * the merge phase means ascending k -- the order the HIP path reproduces. */
static void OSP_FN(coo_sort)(OSP_FN(osp_coo) * a, size_t n, int by_col_row) {
    if (n < 2) return;
    OSP_FN(osp_coo) *tmp = (OSP_FN(osp_coo) *)malloc(n * sizeof(*tmp));
    OSP_FN(osp_coo) *src = a, *dst = tmp;
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n;
            size_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                int take_right = by_col_row ? OSP_FN(coo_less_cr)(&src[j], &src[i])
                                            : OSP_FN(coo_less_rc)(&src[j], &src[i]);
                dst[o++] = take_right ? src[j++] : src[i++];
            }
            while (i < mid) dst[o++] = src[i++];
            while (j < hi) dst[o++] = src[j++];
        }
        OSP_FN(osp_coo) *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(*a));
    free(tmp);
}

/*
 * COO -> compressed (CSR, or CSC when transpose != 0).
 * Reference: coo2csr<transpose>, SimSpGEMM.cpp:102-152, with dupcheck :43-53.
 *   - sort by (row,col) or (col,row)                              :111-121
 *   - adjacent duplicate coordinate -> error 233 (ref: throw(233)) :43-53,:123
 *   - pos[] by run length                                          :125-141
 * DOCUMENTED DIVERGENCE: the reference back-fills every trailing pos==0 with nnz
 * (:143-148), which also overwrites pos[0..] when all non-zeros sit in segment 0
 * and turns such a matrix into an empty one.  That is a bug, not behaviour to
 * keep; the oracle (and the product) fill pos[] exactly.
 * Returns 0, or OSP_ORACLE_ERR_DUPLICATE (233).
 */
int OSP_FN(osp_oracle_coo2csr)(int transpose, size_t nseg, size_t nnz, const uint32_t *row,
                               const uint32_t *col, const OSP_T *val, int64_t *pos /*nseg+1*/,
                               uint32_t *idx /*nnz*/, OSP_T *out_val /*nnz*/) {
    OSP_FN(osp_coo) *coo = (OSP_FN(osp_coo) *)malloc((nnz ? nnz : 1) * sizeof(*coo));
    for (size_t i = 0; i < nnz; i++) {
        coo[i].row = row[i];
        coo[i].col = col[i];
        coo[i].val = val[i];
    }
    OSP_FN(coo_sort)(coo, nnz, transpose);
    for (size_t i = 0; i + 1 < nnz; i++) { /* dupcheck, :43-53 */
        if (coo[i].row == coo[i + 1].row && coo[i].col == coo[i + 1].col) {
            free(coo);
            return OSP_ORACLE_ERR_DUPLICATE;
        }
    }
    for (size_t s = 0; s <= nseg; s++) pos[s] = 0;
    for (size_t i = 0; i < nnz; i++) {
        uint32_t seg = transpose ? coo[i].col : coo[i].row;
        uint32_t in = transpose ? coo[i].row : coo[i].col;
        if (seg >= nseg) {
            free(coo);
            return OSP_ORACLE_ERR_RANGE;
        }
        pos[seg + 1]++;
        idx[i] = in;
        out_val[i] = coo[i].val;
    }
    for (size_t s = 0; s < nseg; s++) pos[s + 1] += pos[s];
    free(coo);
    return 0;
}

/*
 * P = sum_k nnz(A[:,k]) * nnz(B[k,:])   over k in [k0,k1).
 * Reference: mulflops_ref, SimSpGEMM.cpp:884-891.
 */
uint64_t OSP_FN(osp_oracle_mulflops)(size_t k0, size_t k1, const int64_t *a_colptr,
                                     const int64_t *b_rowptr) {
    uint64_t p = 0;
    for (size_t k = k0; k < k1; k++)
        p += (uint64_t)(a_colptr[k + 1] - a_colptr[k]) * (uint64_t)(b_rowptr[k + 1] - b_rowptr[k]);
    return p;
}

/*
 * MULTIPLY phase over the k-slab [k0,k1).
 * Reference: cscMulcsr, SimSpGEMM.cpp:265-281 -- for every k with both A[:,k]
 * and B[k,:] non-empty, for j in A[:,k], for l in B[k,:]:
 *     emit { A.idx_j, B.idx_l, A.val_j * B.val_l }              (:276)
 * The reference returns one COOMatrix per active k; concatenated in k order they
 * are exactly this flat array (order: k asc, A-row asc, B-col asc).  The product
 * is rounded to OSP_T, as value_t arithmetic does in the reference.
 * `out` must hold osp_oracle_mulflops(k0,k1,...) entries.  Returns that count.
 */
static uint64_t OSP_FN(csc_mul_csr)(size_t k0, size_t k1, const int64_t *a_colptr,
                                    const uint32_t *a_rowidx, const OSP_T *a_val,
                                    const int64_t *b_rowptr, const uint32_t *b_colidx,
                                    const OSP_T *b_val, OSP_FN(osp_coo) * out) {
    uint64_t n = 0;
    for (size_t k = k0; k < k1; k++) {
        if (a_colptr[k] == a_colptr[k + 1] || b_rowptr[k] == b_rowptr[k + 1]) continue; /* :271 */
        for (int64_t j = a_colptr[k]; j < a_colptr[k + 1]; j++)
            for (int64_t l = b_rowptr[k]; l < b_rowptr[k + 1]; l++) {
                out[n].row = a_rowidx[j];
                out[n].col = b_colidx[l];
                out[n].val = a_val[j] * b_val[l];
                n++;
            }
    }
    return n;
}

/*
 * Whole path: C = A(CSC, MxK) * B(CSR, KxN) restricted to k in [k0,k1), CSR out.
 *   multiply   cscMulcsr                          SimSpGEMM.cpp:265-281
 *   merge      deduplicateCOO (sort + linear sum) SimSpGEMM.cpp:519-535
 *              - entries whose sum cancels to 0 are KEPT (:529-532 never drops)
 *   CSR build  rowptr by run length, as coo2csr does (:125-141)
 * c_rowptr must hold M+1 entries; *c_colidx / *c_val are malloc'ed here (free
 * with osp_oracle_free).  secs[0..1] (optional) = multiply / merge wall seconds.
 * Returns 0 or an OSP_ORACLE_ERR_* code.
 */
int OSP_FN(osp_oracle_spgemm)(size_t M, size_t K, size_t N, size_t k0, size_t k1,
                              const int64_t *a_colptr, const uint32_t *a_rowidx,
                              const OSP_T *a_val, const int64_t *b_rowptr,
                              const uint32_t *b_colidx, const OSP_T *b_val, int64_t *c_rowptr,
                              uint32_t **c_colidx, OSP_T **c_val, uint64_t *partials,
                              double *secs) {
    (void)N;
    if (k1 > K || k0 > k1) return OSP_ORACLE_ERR_RANGE;
    double t0 = osp_oracle_now();
    uint64_t P = OSP_FN(osp_oracle_mulflops)(k0, k1, a_colptr, b_rowptr);
    OSP_FN(osp_coo) *parts = (OSP_FN(osp_coo) *)malloc((P ? P : 1) * sizeof(*parts));
    if (!parts) return OSP_ORACLE_ERR_ALLOC;
    OSP_FN(csc_mul_csr)(k0, k1, a_colptr, a_rowidx, a_val, b_rowptr, b_colidx, b_val, parts);
    double t1 = osp_oracle_now();

    OSP_FN(coo_sort)(parts, P, 0); /* deduplicateCOO :521 (stable here, see above) */
    uint64_t nnzc = 0;
    for (uint64_t i = 0; i < P; i++) { /* :526-532 */
        if (i == 0 || parts[i].row != parts[i - 1].row || parts[i].col != parts[i - 1].col)
            parts[nnzc++] = parts[i];
        else
            parts[nnzc - 1].val += parts[i].val;
    }
    uint32_t *cc = (uint32_t *)malloc((nnzc ? nnzc : 1) * sizeof(uint32_t));
    OSP_T *cv = (OSP_T *)malloc((nnzc ? nnzc : 1) * sizeof(OSP_T));
    if (!cc || !cv) {
        free(parts); free(cc); free(cv);
        return OSP_ORACLE_ERR_ALLOC;
    }
    for (size_t r = 0; r <= M; r++) c_rowptr[r] = 0;
    for (uint64_t i = 0; i < nnzc; i++) {
        if (parts[i].row >= M) {
            free(parts); free(cc); free(cv);
            return OSP_ORACLE_ERR_RANGE;
        }
        c_rowptr[parts[i].row + 1]++;
        cc[i] = parts[i].col;
        cv[i] = parts[i].val;
    }
    for (size_t r = 0; r < M; r++) c_rowptr[r + 1] += c_rowptr[r];
    double t2 = osp_oracle_now();
    free(parts);
    *c_colidx = cc;
    *c_val = cv;
    if (partials) *partials = P;
    if (secs) {
        secs[0] = t1 - t0;
        secs[1] = t2 - t1;
    }
    return 0;
}

#undef OSP_FN
#undef OSP_CAT
#undef OSP_CAT_
