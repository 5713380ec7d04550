"""ctypes binding of the C ABI declared in ``include/outerspace_spgemm.h``.

The shared library is built in-tree (``outerspace_amd/libouterspace_spgemm.so``) by
``make -C outerspace_amd/csrc`` / ``__graft_entry__.build()``.  There is no Python or CPU
fallback: a missing library is an ImportError, a missing GPU is an ``OspError`` at context
creation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSP_LIBRARY=<path> loads another build of the same ABI (`make debug` / `make ballot` write files of their own, so the
# default name is always the product build)
LIB_PATH = os.environ.get("OSP_LIBRARY") or os.path.join(_HERE, "libouterspace_spgemm.so")

OSP_F32, OSP_F64 = 0, 1
OSP_HOST, OSP_DEVICE = 0, 1
OSP_OK = 0
ERR_DIM, ERR_ARG, ERR_ALLOC, ERR_HIP, ERR_IO, ERR_RANGE, ERR_CAPACITY, ERR_UNSORTED = 1, 2, 3, 4, 5, 6, 7, 8
ERR_DUPLICATE = 233  # reference: throw(233), simulator/SimSpGEMM.cpp:49


class OspError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[osp status {status}] {message}")
        self.status = status


class Config(C.Structure):
    _fields_ = [("validate", C.c_int), ("partial_capacity", C.c_uint64), ("k_begin", C.c_uint64),
                ("k_end", C.c_uint64), ("row_shard_index", C.c_int), ("row_shard_count", C.c_int), ("algorithm", C.c_int),
                ("reserved", C.c_int * 5)]


class ResultInfo(C.Structure):
    _fields_ = [("M", C.c_uint64), ("K", C.c_uint64), ("N", C.c_uint64),
                ("row_begin", C.c_uint64), ("row_end", C.c_uint64),
                ("nnz_a", C.c_uint64), ("nnz_b", C.c_uint64), ("nnz_c", C.c_uint64),
                ("partials", C.c_uint64), ("panels", C.c_uint32), ("light_tiles", C.c_uint64),
                ("heavy_rows", C.c_uint64), ("heavy_partials", C.c_uint64),
                ("sorted_segments", C.c_uint64), ("sorted_partials", C.c_uint64),
                ("ms_symbolic", C.c_float), ("ms_multiply", C.c_float), ("ms_merge", C.c_float),
                ("ms_compact", C.c_float), ("ms_total", C.c_float),
                ("ms_multiply_kernel", C.c_float), ("ms_merge_kernel", C.c_float), ("ms_ingest", C.c_float),
                ("multiply_launches", C.c_uint32), ("merge_launches", C.c_uint32), ("dtype", C.c_int),
                ("ms_split_kernel", C.c_float), ("split_launches", C.c_uint32), ("split_partials", C.c_uint64),
                ("dense_segments", C.c_uint64), ("direct_rows", C.c_uint64), ("direct_partials", C.c_uint64),
                ("ms_direct_plan_kernel", C.c_float), ("direct_plan_launches", C.c_uint32),
                ("rank_atomic", C.c_uint32), ("dense_atomic", C.c_uint32), ("hub_rows", C.c_uint64), ("hub_partials", C.c_uint64), ("hub_cells", C.c_uint64),
                ("ms_hub_plan_kernel", C.c_float), ("hub_plan_launches", C.c_uint32),
                ("output_slack_bytes", C.c_uint64), ("plans_overlapped", C.c_uint64),
                ("gathered_rows", C.c_uint64), ("gathered_partials", C.c_uint64), ("gathered_runs", C.c_uint64),
                ("gathered_short_partials", C.c_uint64), ("ms_expand_kernel", C.c_float), ("expand_launches", C.c_uint32),
                ("expand_partials", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class Panel(C.Structure):
    """osp_panel_t: one finished row panel of a streamed product (device pointers)."""
    _fields_ = [("row_begin", C.c_uint64), ("row_end", C.c_uint64), ("nnz", C.c_uint64),
                ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("vals", C.c_void_p),
                ("index", C.c_uint32), ("count", C.c_uint32), ("reserved", C.c_uint32 * 2)]


PANEL_FN = C.CFUNCTYPE(C.c_int, C.POINTER(Panel), C.c_void_p)

MULTI_MAX_RANKS = 16


class MultiRankInfo(C.Structure):
    """osp_multi_rank_info_t"""
    _fields_ = [("device", C.c_int), ("k_begin", C.c_uint64), ("k_end", C.c_uint64), ("row_begin", C.c_uint64), ("row_end", C.c_uint64),
                ("partials_local", C.c_uint64), ("records_received", C.c_uint64), ("bytes_sent", C.c_uint64), ("nnz_c", C.c_uint64),
                ("ms_symbolic", C.c_float), ("ms_multiply_kernel", C.c_float), ("ms_merge", C.c_float), ("ms_total", C.c_float),
                ("bytes_to", C.c_uint64 * MULTI_MAX_RANKS), ("copy_streams", C.c_int), ("max_copies_outstanding", C.c_int),
                ("max_copies_in_flight", C.c_int), ("ms_exchange", C.c_float)]


class MultiInfo(C.Structure):
    """osp_multi_info_t"""
    _fields_ = [("nranks", C.c_int), ("subpanels", C.c_int), ("M", C.c_uint64), ("K", C.c_uint64), ("N", C.c_uint64),
                ("nnz_c", C.c_uint64), ("partials", C.c_uint64), ("bytes_exchanged", C.c_uint64), ("ms_total", C.c_float),
                ("ms_upload", C.c_float), ("rank", MultiRankInfo * MULTI_MAX_RANKS)]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_ if name != "rank"}
        d["ranks"] = [{n: (list(getattr(self.rank[g], n))[:self.nranks] if n == "bytes_to" else getattr(self.rank[g], n))
                       for n, _ in MultiRankInfo._fields_} for g in range(self.nranks)]
        return d


# every symbol include/outerspace_spgemm.h declares
EXPORTS = [
    "osp_context_create", "osp_context_create_on_stream", "osp_context_destroy", "osp_context_trim",
    "osp_config_default", "osp_last_error_string", "osp_status_string", "osp_spgemm_csc_csr", "osp_spgemm_csc_csr_panels",
    "osp_spgemm_coo", "osp_spgemm_csc_csr_aos", "osp_context_alloc", "osp_context_free",
    "osp_spgemm_partials", "osp_result_partials", "osp_merge_record_parts",
    "osp_merge_csr_parts", "osp_result_info", "osp_result_copy_csr", "osp_result_device_ptrs",
    "osp_result_destroy", "osp_mtx_read", "osp_host_free", "osp_coo_to_compressed_f32",
    "osp_coo_to_compressed_f64", "osp_spgemm_mtx", "osp_result_write_mtx", "osp_csr_bias_relu", "osp_result_coo_rows",
    "osp_stream_copy_probe",
    "osp_multi_context_create", "osp_multi_context_destroy", "osp_multi_operands_create", "osp_multi_operands_destroy", "osp_spgemm_multi",
    "osp_spgemm_csc_csr_multi", "osp_multi_result_info", "osp_multi_result_shard", "osp_multi_result_copy_csr", "osp_multi_result_destroy",
]

_lib = None


def lib():
    """Load the library (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C outerspace_amd/csrc` "
                          "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                          "There is no CPU fallback.")
    # One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7, asked for as
    # "libamdhip64.so" by torch's libraries): loaded first, it also satisfies this library's NEEDED entry; loaded second,
    # the process ends up with two runtimes and torch reports "No HIP GPUs are available".  The package uses torch for
    # device memory anyway, so it goes first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.osp_last_error_string.restype = C.c_char_p
    L.osp_status_string.restype = C.c_char_p
    L.osp_status_string.argtypes = [i32]
    L.osp_context_create.argtypes = [i32, C.POINTER(vp)]
    L.osp_context_create_on_stream.argtypes = [i32, vp, C.POINTER(vp)]
    L.osp_context_destroy.argtypes = [vp]
    L.osp_context_trim.argtypes = [vp]
    L.osp_context_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.osp_context_free.argtypes = [vp, vp]
    L.osp_spgemm_csc_csr_aos.argtypes = [vp, i32, u64, u64, u64, vp, vp, vp, vp, i32, C.POINTER(Config), C.POINTER(vp)]
    L.osp_config_default.argtypes = [C.POINTER(Config)]
    L.osp_config_default.restype = None
    L.osp_spgemm_csc_csr.argtypes = [vp, i32, u64, u64, u64, vp, vp, vp, vp, vp, vp, i32,
                                     C.POINTER(Config), C.POINTER(vp)]
    L.osp_spgemm_csc_csr_panels.argtypes = [vp, i32, u64, u64, u64, vp, vp, vp, vp, vp, vp, i32, C.POINTER(Config), PANEL_FN, vp,
                                            C.POINTER(ResultInfo)]
    L.osp_spgemm_coo.argtypes = [vp, i32, u64, u64, u64, u64, vp, vp, vp, u64, vp, vp, vp, i32, C.POINTER(Config), C.POINTER(vp)]
    L.osp_spgemm_partials.argtypes = [vp, i32, u64, u64, u64, vp, vp, vp, vp, vp, vp, i32, C.POINTER(Config), C.POINTER(vp)]
    L.osp_result_partials.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.osp_merge_record_parts.argtypes = [vp, i32, u64, u64, i32, C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(Config), C.POINTER(vp)]
    L.osp_merge_csr_parts.argtypes = [vp, i32, u64, u64, i32, C.POINTER(vp), C.POINTER(vp),
                                      C.POINTER(vp), i32, C.POINTER(Config), C.POINTER(vp)]
    L.osp_result_info.argtypes = [vp, C.POINTER(ResultInfo)]
    L.osp_result_copy_csr.argtypes = [vp, vp, vp, vp, i32]
    L.osp_result_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.osp_result_destroy.argtypes = [vp]
    L.osp_mtx_read.argtypes = [C.c_char_p, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64),
                               C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.osp_host_free.argtypes = [vp]
    L.osp_host_free.restype = None
    for sfx in ("f32", "f64"):
        getattr(L, f"osp_coo_to_compressed_{sfx}").argtypes = [i32, u64, u64, vp, vp, vp, vp, vp, vp]
    L.osp_spgemm_mtx.argtypes = [vp, i32, C.c_char_p, C.c_char_p, i32, C.POINTER(Config), C.POINTER(vp)]
    L.osp_result_write_mtx.argtypes = [vp, C.c_char_p]
    L.osp_stream_copy_probe.argtypes = [vp, u64, i32, C.POINTER(C.c_double)]
    L.osp_multi_context_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    L.osp_multi_context_destroy.argtypes = [vp]
    L.osp_multi_operands_create.argtypes = [vp, i32, u64, u64, u64, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.osp_multi_operands_destroy.argtypes = [vp]
    L.osp_spgemm_multi.argtypes = [vp, vp, C.POINTER(Config), C.POINTER(vp)]
    L.osp_spgemm_csc_csr_multi.argtypes = [C.POINTER(i32), i32, i32, u64, u64, u64, vp, vp, vp, vp, vp, vp, C.POINTER(Config), C.POINTER(vp),
                                           C.POINTER(vp)]
    L.osp_multi_result_info.argtypes = [vp, C.POINTER(MultiInfo)]
    L.osp_multi_result_shard.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(vp)]
    L.osp_multi_result_copy_csr.argtypes = [vp, vp, vp, vp]
    L.osp_multi_result_destroy.argtypes = [vp]
    L.osp_csr_bias_relu.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.osp_result_coo_rows.argtypes = [vp, vp]
    _lib = L
    return L


def check(status):
    if status != OSP_OK:
        raise OspError(status, lib().osp_last_error_string().decode(errors="replace"))
