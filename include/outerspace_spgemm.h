/*
 * outerspace_spgemm.h -- C ABI of the MI355X-native outer-product SpGEMM.
 *
 * This is the drop-in boundary for the numeric SpGEMM path of anneouyang/OuterSPACE
 * (all citations relative to the reference's simulator/ directory):
 *
 *   reference interface                                         replaced by
 *   ----------------------------------------------------------  -----------------------------
 *   CSRMatrix{pos,data{idx,val}}            common.h:10-16,39-47  SoA arrays (ptr/idx/val); the reference's own AoS
 *                                                                 layout is taken as it stands by osp_spgemm_csc_csr_aos
 *   std::vector<COOMatrix> cscMulcsr(csc, csr)
 *                                           SimSpGEMM.cpp:265-281  osp_spgemm_csc_csr (multiply: inside the merge kernel for the rows
 *                                                                 a plan covers, multiply_kernel / expand_rows_kernel for the rest)
 *   COOMatrix deduplicateCOO(coo)           SimSpGEMM.cpp:519-535  osp_spgemm_csc_csr (merge)
 *   mulflops_ref                            SimSpGEMM.cpp:884-891  osp_result_info.partials
 *   COOMatrix readcoo(istream&, NRow, NCol, sym)
 *                                           SimSpGEMM.cpp:55-100   osp_mtx_read
 *   CSRMatrix coo2csr<transpose>(coo, N)    SimSpGEMM.cpp:102-152  osp_coo_to_compressed_* (host), osp_spgemm_coo (GPU)
 *   dupcheck -> throw(233)                  SimSpGEMM.cpp:43-53    OSP_ERR_DUPLICATE (= 233)
 *   assert(csc.pos.size()==csr.pos.size())  SimSpGEMM.cpp:267,882  OSP_ERR_DIM
 *   main(argv[1]=A.mtx, argv[2]=B.mtx)      SimSpGEMM.cpp:819-894  osp_spgemm_mtx / outerspace_amd/osp_spgemm
 *   x = relu(fc(x)) between two layers      NN_models/models.py:17-31  osp_csr_bias_relu (+ osp_result_coo_rows: the next product's operand)
 *   (nothing: one process, one thread)      SURVEY.md 8e          osp_multi_* -- the k-sharded product over the GPUs of a node
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the ABI; every function
 * returns an osp_status_t (0 = ok); osp_last_error_string() describes the last failure on the
 * calling thread.  Index type u32 (reference index_t, common.h:7), offsets int64 (reference
 * size_t), values f32 or f64 (reference value_t is float, common.h:8; f64 is the north-star
 * target).  Within every compressed segment indices must be strictly ascending (what
 * coo2csr + dupcheck guarantee in the reference).
 *
 * A context owns one GPU, one HIP stream and a buffer pool; it is not thread-safe, use one per
 * thread.  Results are library-owned handles: query, copy out, destroy.
 */
#ifndef OUTERSPACE_SPGEMM_H
#define OUTERSPACE_SPGEMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSP_VERSION 5   /* round of the build: 3 = osp_multi_*, osp_csr_bias_relu, osp_result_coo_rows, direct-row counters;
                            4 = per-destination exchange streams (osp_multi_rank_info_t grew);
                            5 = osp_result_info_t grew (gathered_*, expand_*): since this version the two reference functions
                                of osp_spgemm_csc_csr are ONE kernel for most rows -- the merge forms the partial products
                                cscMulcsr would stage (same products, same order, same bits) */

typedef enum osp_status {
    OSP_OK = 0,
    OSP_ERR_DIM = 1,         /* inner dimensions differ (reference: assert, SimSpGEMM.cpp:267,882) */
    OSP_ERR_ARG = 2,         /* null pointer / bad enum / bad size */
    OSP_ERR_ALLOC = 3,       /* host or device allocation failed */
    OSP_ERR_HIP = 4,         /* HIP runtime error */
    OSP_ERR_IO = 5,          /* file could not be read / written */
    OSP_ERR_RANGE = 6,       /* an index is outside its dimension */
    OSP_ERR_CAPACITY = 7,    /* one output row's partial products exceed the staging capacity */
    OSP_ERR_UNSORTED = 8,    /* indices inside a segment are not ascending */
    OSP_ERR_DUPLICATE = 233  /* duplicate coordinate (reference: throw(233), SimSpGEMM.cpp:49) */
} osp_status_t;

typedef enum osp_dtype { OSP_F32 = 0, OSP_F64 = 1 } osp_dtype_t;
typedef enum osp_memspace { OSP_HOST = 0, OSP_DEVICE = 1 } osp_memspace_t;

/* Two formulations of the same product (identical results, bit for bit):
 *   OSP_ALGO_OUTER    outer products, staged in HBM and merged (the reference's algorithm, SimSpGEMM.cpp:265-297)
 *   OSP_ALGO_ROWWISE  output rows whose partial products fit one merge tile are formed row by row inside the merge
 *                     kernel and never staged (the reference's row-wise alternative, SimSpGEMM.cpp:247-263); longer
 *                     rows still take the outer-product path */
typedef enum osp_algorithm { OSP_ALGO_OUTER = 0, OSP_ALGO_ROWWISE = 1 } osp_algorithm_t;

typedef struct osp_context_s *osp_context_t;
typedef struct osp_result_s *osp_result_t;

/* Tunables of one multiply.  Zero-initialise, then osp_config_default(). */
typedef struct osp_config {
    int validate;               /* 1: check ranges / ordering / duplicates on the device first */
    uint64_t partial_capacity;  /* max partial products staged in HBM at once (0 = auto);
                                   the product is processed in output-row panels of at most this */
    uint64_t k_begin, k_end;    /* restrict the shared dimension to [k_begin,k_end); k_end = 0
                                   means K.  This is the k-sharded multi-GPU mode. */
    int row_shard_index;        /* row-sharded multi-GPU mode: compute only output-row range number        */
    int row_shard_count;        /* `index` of `count` ranges balanced by partial products (count <= 1: all  */
                                /* rows).  Every rank derives the same ranges from the operands alone: no   */
                                /* communication.  The result then has row_end - row_begin rows.            */
    int algorithm;              /* osp_algorithm_t (0 = outer product) */
    int reserved[5];
} osp_config_t;

/* What one multiply did.  Times are device milliseconds measured with HIP events on the
 * context's stream. */
typedef struct osp_result_info {
    uint64_t M, K, N;           /* M = rows of the RESULT (the shard's rows in row-sharded mode) */
    uint64_t row_begin, row_end;/* output rows this result covers */
    uint64_t nnz_a, nnz_b, nnz_c;
    uint64_t partials;          /* P = sum_k nnz(A[:,k]) * nnz(B[k,:])  (reference mulflops_ref) */
    uint32_t panels;            /* output-row panels processed */
    uint64_t light_tiles;       /* merge tiles reduced in LDS */
    uint64_t heavy_rows;        /* rows longer than one LDS tile: split by column range first */
    uint64_t heavy_partials;    /* partial products in those rows */
    uint64_t sorted_segments;   /* segments still too long after the split: global-sort path */
    uint64_t sorted_partials;
    float ms_symbolic, ms_multiply, ms_merge, ms_compact, ms_total;   /* phases (all launches in them) */
    float ms_multiply_kernel, ms_merge_kernel;  /* multiply_kernel / merge_tiles_kernel launches alone */
    float ms_ingest;            /* osp_spgemm_coo / osp_spgemm_mtx: COO -> CSC/CSR on the device (included in ms_total) */
    uint32_t multiply_launches, merge_launches; /* number of those launches */
    int dtype;
    float ms_split_kernel;      /* split_row_kernel launches alone (long rows of up to 64 K partial products) */
    uint32_t split_launches;
    uint64_t split_partials;    /* partial products those launches moved (lower bound: heavy_partials minus the
                                   capacity of the stretch-split jobs) */
    uint64_t dense_segments;    /* over-long segments of hub rows reduced by dense accumulation (no sort) */
    uint64_t direct_rows;       /* long rows the multiply phase wrote straight into their column ranges (no split pass) */
    uint64_t direct_partials;   /* partial products in those rows */
    float ms_direct_plan_kernel;/* direct_plan_kernel launches alone (range tables and cells of those rows) */
    uint32_t direct_plan_launches;
    uint32_t rank_atomic;       /* 1: stable radix ranks from the return order of LDS atomics, 0: from ballot matching   */
    uint32_t dense_atomic;      /* 1: dense segments summed by LDS floating-point atomics, 0: by ballot ranks and rounds */
                                /* (both variants are exact; a context picks them by a self-test)         */
    uint64_t hub_rows;          /* since version 4: rows beyond the one-workgroup planner (> 128 K partial products) that the multiply
                                   wrote straight into uniform column blocks (no stretch split) */
    uint64_t hub_partials;      /* partial products in those rows */
    uint64_t hub_cells;         /* (chunk, run of B's row) cells planned for them: one per run the multiply writes */
    float ms_hub_plan_kernel;   /* the two hub_plan_kernel launches, the scans between them and the read-back, per panel with hub rows */
    uint32_t hub_plan_launches; /* panels with hub rows */
    uint64_t output_slack_bytes;/* since version 4: bytes of the result's colidx / vals allocations beyond nnz_c entries (they are
                                   sized by the bound sum_i min(U_i, N) before the merge; copied to exact size only when BOTH hold:
                                   the exact-size allocation would be under 70 % of the bound-sized one, and the slack exceeds a
                                   tenth of the device's memory (per result: several live results each keep theirs) -- 0 after
                                   such a copy) */
    uint64_t plans_overlapped;  /* since version 4: panels whose plan ran on the context's second stream, beside the multiply of the
                                   panel before (products of several panels; the phase and kernel times of the two then overlap) */
    uint64_t gathered_rows;     /* since version 5: direct rows that were never written: the merge kernel formed their partial
                                   products itself, tile by tile, from the plan's run descriptors (cscMulcsr inside the merge) */
    uint64_t gathered_partials; /* partial products in those rows */
    uint64_t gathered_runs;     /* run descriptors planned for them: one per non-empty (chunk, column range) cell */
    uint64_t gathered_short_partials; /* partial products of the rows that fit a merge tile, formed the same way (their runs are
                                   their chunks); 0: those rows were staged by the multiply phase */
    float ms_expand_kernel;     /* since version 5: expand_rows_kernel launches alone -- the long rows beyond the planner, staged row by
                                   row from the chunk table when everything else is gathered */
    uint32_t expand_launches;
    uint64_t expand_partials;   /* partial products those launches wrote */
} osp_result_info_t;

/* ---- context ------------------------------------------------------------------------- */
int osp_context_create(int device, osp_context_t *ctx);
/* Same, but all work is enqueued on an existing hipStream_t (e.g. torch's current stream).  (Products of several panels run
   part of their planning on a second stream of the context's own, forked from and joined into this one by events inside the
   call: to the caller the call's work is ordered on the given stream, before and after.) */
int osp_context_create_on_stream(int device, void *hip_stream, osp_context_t *ctx);
int osp_context_destroy(osp_context_t ctx);
/* Return pooled device memory to the driver. */
int osp_context_trim(osp_context_t ctx);
/* Device memory from / back to the context's buffer pool (the same pool the products take their staging buffers from).
 * The multi-GPU exchange receives into such buffers, so that the memory a finished local product has just released is
 * reused instead of being allocated a second time next to the pool.  No reference counterpart. */
int osp_context_alloc(osp_context_t ctx, uint64_t bytes, void **device_ptr);
int osp_context_free(osp_context_t ctx, void *device_ptr);
void osp_config_default(osp_config_t *cfg);
const char *osp_last_error_string(void);
const char *osp_status_string(int status);

/* ---- the hot path --------------------------------------------------------------------- */
/*
 * C (MxN, CSR) = A (MxK, CSC) * B (KxN, CSR).
 *   a_colptr[K+1], a_rowidx[nnzA], a_vals[nnzA]   -- reference `csc`  (cscMulcsr arg 1)
 *   b_rowptr[K+1], b_colidx[nnzB], b_vals[nnzB]   -- reference `csr`  (cscMulcsr arg 2)
 * `space` says where ALL six input arrays live.  cfg may be NULL (defaults).
 * Replaces cscMulcsr (SimSpGEMM.cpp:265-281) + deduplicateCOO (:519-535); the result has
 * ascending column indices per row and keeps entries that cancel to zero, as the reference does.
 */
int osp_spgemm_csc_csr(osp_context_t ctx, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                       const int64_t *a_colptr, const uint32_t *a_rowidx, const void *a_vals,
                       const int64_t *b_rowptr, const uint32_t *b_colidx, const void *b_vals,
                       osp_memspace_t space, const osp_config_t *cfg, osp_result_t *result);

/*
 * The same product on the reference's IN-MEMORY layout, with no conversion on the caller's side:
 *   CSRMatrix{ std::vector<size_t> pos; std::vector<CSRElement{index_t idx; value_t val}> data; }   common.h:10-16,39-47
 * a_pos / b_pos = csc.pos.data() / csr.pos.data() (K+1 offsets of 64 bits, size_t on every LP64 host), a_data / b_data
 * = csc.data.data() / csr.data.data(): PACKED records {u32 idx; T val} -- 8 bytes for value_t = float (the reference as
 * it ships, common.h:8), 12 bytes for double (`#pragma pack(push, 1)`, common.h:10).  The records are copied to the
 * device as they are and split into index / value arrays there; everything else is osp_spgemm_csc_csr.
 * Drop-in for `cscMulcsr(csc, csr)` (SimSpGEMM.cpp:265) + deduplicateCOO (:519).
 */
int osp_spgemm_csc_csr_aos(osp_context_t ctx, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                           const uint64_t *a_pos, const void *a_data, const uint64_t *b_pos, const void *b_data,
                           osp_memspace_t space, const osp_config_t *cfg, osp_result_t *result);

/*
 * The same product, STREAMED: output rows are merged in consecutive row panels (the library's unit of work when the
 * partial products exceed the staging memory) and every finished panel is handed to `fn` at once; its buffers are
 * reused for the next panel, so C is never resident as a whole -- the only way to run products whose result does not
 * fit the GPU (R-MAT-22 with Graph500 parameters has nnzC ~ 7e10).  Inside the callback the panel's
 * arrays are complete device memory (nothing pending on them); work the callback queues on the context's stream
 * is waited for before the buffers are reused, anything else must be finished before it returns.  A non-zero return
 * aborts the product with OSP_ERR_ARG.  Panels arrive in row order and tile [0, M) (or the row shard) exactly.
 * `info` (may be NULL) receives the same counters osp_result_info reports.  No reference counterpart: the reference
 * materialises all partial products and the whole result in host vectors (SimSpGEMM.cpp:265-281).
 */
typedef struct osp_panel {
    uint64_t row_begin, row_end; /* output rows [row_begin, row_end) */
    uint64_t nnz;                /* entries of the panel */
    const int64_t *rowptr;       /* device, row_end - row_begin + 1 offsets into colidx/vals, rowptr[0] = 0 */
    const uint32_t *colidx;      /* device, ascending inside a row */
    const void *vals;            /* device, dtype of the call */
    uint32_t index, count;       /* this is panel `index` of `count` */
    uint32_t reserved[2];
} osp_panel_t;
typedef int (*osp_panel_fn)(const osp_panel_t *panel, void *user);
int osp_spgemm_csc_csr_panels(osp_context_t ctx, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                              const int64_t *a_colptr, const uint32_t *a_rowidx, const void *a_vals,
                              const int64_t *b_rowptr, const uint32_t *b_colidx, const void *b_vals,
                              osp_memspace_t space, const osp_config_t *cfg, osp_panel_fn fn, void *user,
                              osp_result_info_t *info);

/*
 * Same product from COO operands in any order: the conversion the reference does on the host with two
 * std::sorts -- csc = coo2csr<true>(A, K), csr = coo2csr(B, K), SimSpGEMM.cpp:878-879 / :102-152 -- runs on the
 * GPU (stable radix sorts by (column,row) and (row,column)).  A duplicate coordinate in either operand returns
 * OSP_ERR_DUPLICATE (233), an index outside its dimension OSP_ERR_RANGE.  `space` covers all six arrays.
 */
int osp_spgemm_coo(osp_context_t ctx, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N, uint64_t nnz_a,
                   const uint32_t *a_rows, const uint32_t *a_cols, const void *a_vals, uint64_t nnz_b,
                   const uint32_t *b_rows, const uint32_t *b_cols, const void *b_vals, osp_memspace_t space,
                   const osp_config_t *cfg, osp_result_t *result);

/*
 * Sum `nparts` CSR matrices of identical shape (MxN) into one CSR: the final step of the
 * k-sharded multi-GPU product after the partial CSRs have been exchanged.
 * rowptrs[p][M+1], colidxs[p][nnz_p], valss[p][nnz_p]; `space` as above.
 * Entries are summed in part order (p ascending).
 */
int osp_merge_csr_parts(osp_context_t ctx, osp_dtype_t dtype, uint64_t M, uint64_t N, int nparts,
                        const int64_t *const *rowptrs, const uint32_t *const *colidxs,
                        const void *const *valss, osp_memspace_t space, const osp_config_t *cfg,
                        osp_result_t *result);

/*
 * The MULTIPLY phase alone -- cscMulcsr (SimSpGEMM.cpp:265-281) without deduplicateCOO: the partial products of
 * C = A * B as they are staged, NOT merged.  Packed records {uint32 col; T val} (8 bytes for f32, 12 for f64; the layout
 * of the reference's CSRElement), grouped by output row: row i owns records [rowptr[i], rowptr[i+1]) in ascending k,
 * columns ascending inside one k; equal columns of different k are still separate records.  This is what a rank of the
 * k-sharded multi-GPU product sends when the product hardly compresses (R-MAT: nnz(C) / P = 0.97): merging locally first
 * would merge everything twice.  The receiver sums the parts with osp_merge_record_parts -- in part order, and inside a
 * part in record order, i.e. in ascending k overall: the same order, and the same bits, as the single-GPU product.
 * All P records are held at once (no row panels): OSP_ERR_ALLOC when they do not fit.
 * osp_result_info: partials = nnz_c = P.  osp_result_copy_csr / osp_result_device_ptrs do not apply to such a result;
 * use osp_result_partials (device pointers, valid until osp_result_destroy).
 */
int osp_spgemm_partials(osp_context_t ctx, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                        const int64_t *a_colptr, const uint32_t *a_rowidx, const void *a_vals,
                        const int64_t *b_rowptr, const uint32_t *b_colidx, const void *b_vals,
                        osp_memspace_t space, const osp_config_t *cfg, osp_result_t *result);
int osp_result_partials(osp_result_t r, const int64_t **rowptr, const void **records);
/*
 * Sum `nparts` collections of records of identical shape (M rows, columns < N) into one CSR: rowptrs[p][M+1] offsets in
 * records, records[p] packed {uint32 col; T val}; rows need be neither sorted nor free of duplicates.
 * Precondition: offsets monotone from 0, every column < N.  With cfg->validate != 0 both are checked on the device first
 * (OSP_ERR_ARG / OSP_ERR_RANGE); without it a violation indexes out of bounds on the device.
 */
int osp_merge_record_parts(osp_context_t ctx, osp_dtype_t dtype, uint64_t M, uint64_t N, int nparts,
                           const int64_t *const *rowptrs, const void *const *records, osp_memspace_t space,
                           const osp_config_t *cfg, osp_result_t *result);

/* ---- several GPUs of one node: the k-sharded product (SURVEY.md 8e / 8b "optional device ordinal list") ------------- */
/*
 * No reference counterpart (the reference is one process, one thread).  k is cut into one slab per rank, balanced by
 * partial products; rank g holds only columns [k_g, k_g+1) of A and rows [k_g, k_g+1) of B and multiplies them; the
 * partial products travel ONCE, unmerged, to the rank that owns their output row (direct GPU-to-GPU copies on one stream per
 * destination -- up to G-1 per rank in flight, one per link -- pipelined behind the multiply sub-panel by sub-panel), and every
 * rank merges its row range, on a stream of its own, as the pieces arrive -- parts in rank order =
 * ascending k, the single-GPU summation order: the result is bit-identical to osp_spgemm_csc_csr.  The result stays
 * row-sharded (shard g on device g); osp_multi_result_copy_csr gathers it to the host.
 * devices[] may name an ordinal more than once: logical ranks that share a GPU (how the tests run on a one-GPU box).
 * The calling thread drives rank 0 and starts one host thread per further rank for the duration of a product.
 */
#define OSP_MULTI_MAX_RANKS 16
typedef struct osp_multi_context_s *osp_multi_context_t;
typedef struct osp_multi_operands_s *osp_multi_operands_t;
typedef struct osp_multi_result_s *osp_multi_result_t;
typedef struct osp_multi_rank_info {
    int device;
    uint64_t k_begin, k_end;        /* the rank's slab of the shared dimension */
    uint64_t row_begin, row_end;    /* the output rows it owns */
    uint64_t partials_local;        /* partial products it formed */
    uint64_t records_received;      /* partial products it merged (its own included) */
    uint64_t bytes_sent;            /* records copied to OTHER ranks */
    uint64_t nnz_c;                 /* entries of its shard */
    float ms_symbolic, ms_multiply_kernel, ms_merge, ms_total;   /* host clock of the rank's thread; kernel time from HIP events */
    /* the exchange (since version 4): every destination has a copy stream of its own, so a rank's copies run side by side */
    uint64_t bytes_to[OSP_MULTI_MAX_RANKS];   /* ... of bytes_sent, per destination rank */
    int copy_streams;               /* exchange streams this rank used: one per destination = one per link */
    int max_copies_outstanding;     /* most peer copies queued on their streams and not complete at one time (sampled by the host
                                       after every sub-panel; each only waits for its own piece to be multiplied) */
    int max_copies_in_flight;       /* most peer copies whose [start, end] overlapped on the device (HIP event timestamps) */
    float ms_exchange;              /* first copy start -> last copy end */
} osp_multi_rank_info_t;
typedef struct osp_multi_info {
    int nranks, subpanels;          /* sub-panels per row range: the granularity of the multiply / copy / merge pipeline */
    uint64_t M, K, N, nnz_c, partials, bytes_exchanged;
    float ms_total;                 /* wall time of the product (operands resident on their ranks) */
    float ms_upload;                /* host -> devices distribution of the slabs (osp_multi_operands_create) */
    osp_multi_rank_info_t rank[OSP_MULTI_MAX_RANKS];
} osp_multi_info_t;
int osp_multi_context_create(const int *devices, int ndev, osp_multi_context_t *mc);
int osp_multi_context_destroy(osp_multi_context_t mc);
/* Cut k into slabs and put every rank's slab on its GPU (HOST operands, the six arrays of osp_spgemm_csc_csr). */
int osp_multi_operands_create(osp_multi_context_t mc, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                              const int64_t *a_colptr, const uint32_t *a_rowidx, const void *a_vals,
                              const int64_t *b_rowptr, const uint32_t *b_colidx, const void *b_vals,
                              osp_multi_operands_t *ops);
int osp_multi_operands_destroy(osp_multi_operands_t ops);
/* The product of resident slabs (repeatable).  cfg: validate and partial_capacity are honoured. */
int osp_spgemm_multi(osp_multi_context_t mc, osp_multi_operands_t ops, const osp_config_t *cfg, osp_multi_result_t *result);
/* All of it in one call -- what a caller of cscMulcsr(csc, csr) (SimSpGEMM.cpp:265) with several GPUs writes.  The context
 * comes back because the result's shards live in its pools: destroy the result first, then the context. */
int osp_spgemm_csc_csr_multi(const int *devices, int ndev, osp_dtype_t dtype, uint64_t M, uint64_t K, uint64_t N,
                             const int64_t *a_colptr, const uint32_t *a_rowidx, const void *a_vals,
                             const int64_t *b_rowptr, const uint32_t *b_colidx, const void *b_vals,
                             const osp_config_t *cfg, osp_multi_context_t *mc, osp_multi_result_t *result);
int osp_multi_result_info(osp_multi_result_t r, osp_multi_info_t *info);
/* Shard `rank` (borrowed handle, valid until osp_multi_result_destroy): rows [row_begin, row_end), rowptr local to it. */
int osp_multi_result_shard(osp_multi_result_t r, int rank, uint64_t *row_begin, uint64_t *row_end, osp_result_t *shard);
/* The whole CSR on the host: rowptr[M+1], colidx[nnz_c], vals[nnz_c]; any pointer may be NULL. */
int osp_multi_result_copy_csr(osp_multi_result_t r, int64_t *rowptr, uint32_t *colidx, void *vals);
int osp_multi_result_destroy(osp_multi_result_t r);

/* ---- between the layers of a sparse MLP --------------------------------- */
/*
 * out = relu(C + bias) with the zeros dropped, as a new CSR result on the same context: what NN_models/models.py:17-31
 * does to a layer's product before it feeds the next layer (x = relu(fc(x)); fc adds its bias to EVERY element, so a
 * biased row is dense before the ReLU: out[i,j] = relu(C[i,j] + bias[j]) for all j, an absent C[i,j] counting as 0).
 * bias: N values of C's dtype (host or device) or NULL (then only stored entries are touched); relu != 0 applies
 * max(., 0).  Row pointers are exact, columns ascend.  `in` stays valid.  No reference kernel: the reference's layers
 * are dense torch ops and its hand-off is a .mtx file per activation (get_mtx_files.py:76-96).
 */
int osp_csr_bias_relu(osp_result_t in, const void *bias, osp_memspace_t bias_space, int relu, osp_result_t *out);
/* The row index of every entry of a CSR result (nnz_c values, DEVICE memory of the caller): with the result's colidx /
 * vals arrays that is the COO form osp_spgemm_coo takes, so an activation feeds the next product without leaving the GPU. */
int osp_result_coo_rows(osp_result_t r, uint32_t *rows_device);

/* ---- measurement aid: what a plain stream reaches on this device ------------------------------------------------------ */
/* A 16-bytes-per-lane copy of `bytes` bytes, `reps` times on the context's stream; *gbps = (read + written) / time in GB/s.
 * No reference counterpart in the product path: SURVEY.md 8d asks for "an achievable-stream number on the box" beside the
 * data-sheet peak (the reference prints its simulated DRAM rate, SimOuterSPACE.cpp:684-686). */
int osp_stream_copy_probe(osp_context_t ctx, uint64_t bytes, int reps, double *gbps);

/* ---- results --------------------------------------------------------------------------- */
int osp_result_info(osp_result_t r, osp_result_info_t *info);
/* Copy the CSR out.  rowptr[M+1], colidx[nnz_c], vals[nnz_c]; any pointer may be NULL. */
int osp_result_copy_csr(osp_result_t r, int64_t *rowptr, uint32_t *colidx, void *vals,
                        osp_memspace_t space);
/* Borrow the device arrays (valid until osp_result_destroy).  They go back to the context's buffer pool then, and the next
 * product of the context may write to them at once, on the CONTEXT's stream: work the caller has queued on a stream of its
 * own that still reads them (a copy, a reduction) must have completed before osp_result_destroy is called. */
int osp_result_device_ptrs(osp_result_t r, const int64_t **rowptr, const uint32_t **colidx,
                           const void **vals);
int osp_result_destroy(osp_result_t r);

/* ---- ingest (host side of the reference CLI) ------------------------------------------ */
/*
 * MatrixMarket coordinate reader with the reference's rules (readcoo, SimSpGEMM.cpp:55-100):
 * lines whose first non-blank character is '%' and blank lines are skipped, the first kept line
 * is "rows cols nnz", entries are 1-based, a missing value means 1.0, `symmetric` mirrors
 * off-diagonal entries.  Values come back as parsed doubles.  Arrays are malloc'ed; release
 * with osp_host_free.
 */
int osp_mtx_read(const char *path, int symmetric, uint64_t *nrow, uint64_t *ncol, uint64_t *nnz,
                 uint32_t **rows, uint32_t **cols, double **vals);
void osp_host_free(void *p);
/*
 * COO -> compressed on the host (coo2csr<transpose>, SimSpGEMM.cpp:102-152): by_col = 0 gives
 * CSR over `nseg` rows, by_col = 1 gives CSC over `nseg` columns.  Duplicate coordinates return
 * OSP_ERR_DUPLICATE (233).  Unlike the reference, a matrix with a single non-empty segment is
 * converted correctly (the reference's back-fill at :143-148 empties it).
 */
int osp_coo_to_compressed_f32(int by_col, uint64_t nseg, uint64_t nnz, const uint32_t *rows,
                              const uint32_t *cols, const float *vals, int64_t *ptr, uint32_t *idx,
                              float *out_vals);
int osp_coo_to_compressed_f64(int by_col, uint64_t nseg, uint64_t nnz, const uint32_t *rows,
                              const uint32_t *cols, const double *vals, int64_t *ptr, uint32_t *idx,
                              double *out_vals);
/*
 * The reference CLI's data flow (main, SimSpGEMM.cpp:819-891) in one call: read both files,
 * transpose the second operand when transpose_b != 0 (the reference always does, :852-856),
 * build CSC(A) / CSR(B'), multiply on the GPU.  *partials receives P ("mul flops ref", :891).
 */
int osp_spgemm_mtx(osp_context_t ctx, osp_dtype_t dtype, const char *path_a, const char *path_b,
                   int transpose_b, const osp_config_t *cfg, osp_result_t *result);
/* Write a CSR result as MatrixMarket "coordinate real general" (1-based, row-major order). */
int osp_result_write_mtx(osp_result_t r, const char *path);

#ifdef __cplusplus
}
#endif
#endif /* OUTERSPACE_SPGEMM_H */
