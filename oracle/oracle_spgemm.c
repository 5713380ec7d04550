/*
 * oracle_spgemm.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's SpGEMM path (anneouyang/OuterSPACE,
 * simulator/SimSpGEMM.cpp + common.h).  It exists to CHECK the HIP product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (outerspace_amd/) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against the
 * fixtures in tests/golden/, which were produced by the reference's own functions
 * compiled from /root/reference (oracle/ref_driver.cpp -> oracle/_ref/, recipe in
 * oracle/Makefile, generator tests/golden/make_golden.py).
 *
 * Build: make -C oracle   ->  oracle/liboracle_spgemm.so
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define OSP_ORACLE_ERR_DUPLICATE 233 /* reference: throw(233), SimSpGEMM.cpp:49 */
#define OSP_ORACLE_ERR_RANGE 2
#define OSP_ORACLE_ERR_ALLOC 3
#define OSP_ORACLE_ERR_IO 4

static double osp_oracle_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void osp_oracle_free(void *p) { free(p); }

/*
 * MatrixMarket coordinate reader.  Reference: readcoo, SimSpGEMM.cpp:55-100.
 *   - a line is skipped when its first non-blank character is '%' or it is blank  :66-77
 *     (blank = only ' ' and '\t'; the banner "%%MatrixMarket ..." is therefore
 *      skipped, never parsed)
 *   - first kept line: "NRow NCol NNZ" (NNZ only sizes the reserve)               :79-88
 *   - each further line: "row col val" 1-based -> 0-based                         :90-94
 *     fewer than 3 fields parsed -> val = 1.0 (pattern files)                     :92-93
 *   - sym != 0 mirrors off-diagonal entries (reference hard-wires false, :821)    :95-96
 * Values are returned as parsed doubles; the caller narrows with (float) exactly
 * as `value_t(val)` does at :94.  Output arrays are malloc'ed (osp_oracle_free).
 */
int osp_oracle_readcoo(const char *path, int sym, uint64_t *nrow, uint64_t *ncol, uint64_t *nnz,
                       uint32_t **rows, uint32_t **cols, double **vals) {
    FILE *f = fopen(path, "r");
    if (!f) return OSP_ORACLE_ERR_IO;
    size_t cap = 1024, n = 0;
    uint32_t *r = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint32_t *c = (uint32_t *)malloc(cap * sizeof(uint32_t));
    double *v = (double *)malloc(cap * sizeof(double));
    char *line = NULL;
    size_t linecap = 0;
    int firstline = 1;
    size_t NRow = 0, NCol = 0, NNZ = 0;
    while (getline(&line, &linecap, f) >= 0) {
        int skip = 1;
        for (const char *p = line; *p && *p != '\n' && *p != '\r'; p++) {
            if (*p == ' ' || *p == '\t') continue;
            if (*p == '%') break;
            skip = 0;
            break;
        }
        if (skip) continue;
        if (firstline) {
            sscanf(line, "%zu %zu %zu", &NRow, &NCol, &NNZ);
            firstline = 0;
            continue;
        }
        size_t row = 0, col = 0;
        double val = 0.0;
        if (sscanf(line, "%zu %zu %lf", &row, &col, &val) < 3) val = 1.0;
        if (n + 2 > cap) {
            cap *= 2;
            r = (uint32_t *)realloc(r, cap * sizeof(uint32_t));
            c = (uint32_t *)realloc(c, cap * sizeof(uint32_t));
            v = (double *)realloc(v, cap * sizeof(double));
        }
        r[n] = (uint32_t)(row - 1); c[n] = (uint32_t)(col - 1); v[n] = val; n++;
        if (sym && row != col) {
            r[n] = (uint32_t)(col - 1); c[n] = (uint32_t)(row - 1); v[n] = val; n++;
        }
    }
    free(line);
    fclose(f);
    *nrow = NRow; *ncol = NCol; *nnz = n;
    *rows = r; *cols = c; *vals = v;
    return 0;
}

#define OSP_T float
#define OSP_SFX f32
#include "oracle_impl.h"
#undef OSP_T
#undef OSP_SFX

#define OSP_T double
#define OSP_SFX f64
#include "oracle_impl.h"
#undef OSP_T
#undef OSP_SFX
