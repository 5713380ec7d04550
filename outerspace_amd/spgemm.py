"""Host-side Python surface of the MI355X outer-product SpGEMM.

Mirrors the reference's call surface for this path:

* ``spgemm_mtx(a, b)``            -- ``./simulator A.mtx B.mtx`` (``simulator/SimSpGEMM.cpp:819-894``):
  two MatrixMarket files in, the second operand transposed by default (``:852-856``), CSR out.
* ``spgemm_csc_csr(...)``         -- ``cscMulcsr(csc, csr)`` + sort/sum (``:265-281``, ``:519-535``).
* ``spgemm(A, B, transpose_b)``   -- the entry point SURVEY.md section 0 places beside
  ``NN_models/sparse_util.py``: dense/scipy operands as ``get_mtx_files.py`` dumps them
  (activation ``batch x in``, ``nn.Linear`` weight ``out x in``) -> ``act @ W.T`` as scipy CSR.
* ``read_mtx`` / ``coo_to_csr`` / ``coo_to_csc`` -- ``readcoo`` (``:55-100``), ``coo2csr`` (``:102-152``).

All numeric work happens in ``libouterspace_spgemm.so`` on the GPU; nothing here computes.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from ._lib import OspError  # noqa: F401  (re-export)

_DT = {np.dtype(np.float32): _lib.OSP_F32, np.dtype(np.float64): _lib.OSP_F64}


def _ptr(a):
    return C.c_void_p(a.ctypes.data if a.size else 0)


class CsrResult:
    """Library-owned CSR result.  ``rowptr``/``colidx``/``vals`` copy to the host on first use."""

    def __init__(self, ctx, handle):
        self._ctx, self._h = ctx, handle
        ctx._results.add(self)  # a context closes its results before it goes away
        info = _lib.ResultInfo()
        _lib.check(_lib.lib().osp_result_info(handle, C.byref(info)))
        self.info = info.as_dict()
        self.shape = (info.M, info.N)
        self.nnz = info.nnz_c
        self.dtype = np.float32 if info.dtype == _lib.OSP_F32 else np.float64
        self._host = None

    def to_host(self):
        if self._host is None:
            rowptr = np.empty(self.shape[0] + 1, np.int64)
            colidx = np.empty(self.nnz, np.uint32)
            vals = np.empty(self.nnz, self.dtype)
            _lib.check(_lib.lib().osp_result_copy_csr(self._h, _ptr(rowptr), _ptr(colidx), _ptr(vals), _lib.OSP_HOST))
            self._host = (rowptr, colidx, vals)
        return self._host

    rowptr = property(lambda self: self.to_host()[0])
    colidx = property(lambda self: self.to_host()[1])
    vals = property(lambda self: self.to_host()[2])

    def partials_ptrs(self):
        """Result of ``spgemm_partials_device``: (rowptr, records) device addresses -- int64 record offsets per row, packed
        ``{u32 col; T val}`` records -- valid until ``close()``."""
        r, rec = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().osp_result_partials(self._h, C.byref(r), C.byref(rec)))
        return r.value or 0, rec.value or 0

    def device_ptrs(self):
        """(rowptr, colidx, vals) device addresses, valid until ``close()``."""
        r, c, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().osp_result_device_ptrs(self._h, C.byref(r), C.byref(c), C.byref(v)))
        return r.value or 0, c.value or 0, v.value or 0

    def bias_relu(self, bias=None, relu=True):
        """``relu(C + bias)`` with the zeros dropped, as a new CSR result on the device (``osp_csr_bias_relu``): what
        ``models.py:17-31`` does between two layers.  bias: N values (numpy, C's dtype) or None."""
        b = None if bias is None else np.ascontiguousarray(bias, self.dtype)
        if b is not None and b.shape != (self.shape[1],):
            raise ValueError(f"bias must have {self.shape[1]} entries")
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_csr_bias_relu(self._h, _ptr(b) if b is not None else None, _lib.OSP_HOST, int(bool(relu)), C.byref(h)))
        return CsrResult(self._ctx, h)

    def coo_rows_into(self, rows_device_ptr):
        """Row index of every entry into caller-owned DEVICE memory (nnz u32 values): with ``device_ptrs()[1:]`` the COO
        form ``Context.spgemm_coo_device`` takes (``osp_result_coo_rows``)."""
        _lib.check(_lib.lib().osp_result_coo_rows(self._h, C.c_void_p(int(rows_device_ptr))))

    def to_scipy(self):
        import scipy.sparse as sp
        rowptr, colidx, vals = self.to_host()
        return sp.csr_matrix((vals, colidx.astype(np.int64), rowptr), shape=self.shape)

    def write_mtx(self, path):
        _lib.check(_lib.lib().osp_result_write_mtx(self._h, os.fsencode(path)))

    def close(self):
        if self._h is not None:
            _lib.lib().osp_result_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


ALGORITHMS = {"outer": 0, "rowwise": 1}  # osp_algorithm_t


def aos_dtype(value_dtype):
    """numpy view of the reference's packed ``CSRElement{index_t idx; value_t val}`` (``common.h:10-16``)."""
    return np.dtype([("idx", "<u4"), ("val", np.dtype(value_dtype).newbyteorder("<"))], align=False)


class Context:
    """One GPU + one HIP stream + a buffer pool (``osp_context_t``)."""

    def __init__(self, device=0, stream=None):
        self._h = None
        self._results = weakref.WeakSet()
        h = C.c_void_p()
        if stream is None:
            _lib.check(_lib.lib().osp_context_create(device, C.byref(h)))
        else:
            _lib.check(_lib.lib().osp_context_create_on_stream(device, C.c_void_p(stream), C.byref(h)))
        self._h = h
        self.device = device
        #: "outer" (default) or "rowwise": which formulation the products of this context use (osp_config_t.algorithm)
        self.algorithm = "outer"

    def close(self):
        if self._h is not None:
            for r in list(self._results):
                r.close()
            _lib.lib().osp_context_destroy(self._h)
            self._h = None

    def trim(self):
        _lib.check(_lib.lib().osp_context_trim(self._h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _config(self, validate, partial_capacity, k_range, row_shard=None):
        cfg = _lib.Config()
        _lib.lib().osp_config_default(C.byref(cfg))
        cfg.validate = int(bool(validate))
        cfg.partial_capacity = int(partial_capacity or 0)
        if k_range is not None:
            cfg.k_begin, cfg.k_end = int(k_range[0]), int(k_range[1])
        if row_shard is not None:
            cfg.row_shard_index, cfg.row_shard_count = int(row_shard[0]), int(row_shard[1])
        cfg.algorithm = ALGORITHMS[self.algorithm]
        return cfg

    def spgemm_csc_csr(self, M, K, N, a_colptr, a_rowidx, a_vals, b_rowptr, b_colidx, b_vals, *,
                       validate=True, partial_capacity=0, k_range=None, row_shard=None):
        """C = A(CSC) * B(CSR) with numpy (host) operands.  row_shard=(i, G): only the i-th of G output-row ranges
        (balanced by partial products; ``result.info['row_begin'/'row_end']`` say which rows came back)."""
        dt = np.dtype(a_vals.dtype)
        if dt not in _DT or np.dtype(b_vals.dtype) != dt:
            raise TypeError("values must both be float32 or both float64")
        arrs = [np.ascontiguousarray(a_colptr, np.int64), np.ascontiguousarray(a_rowidx, np.uint32),
                np.ascontiguousarray(a_vals, dt), np.ascontiguousarray(b_rowptr, np.int64),
                np.ascontiguousarray(b_colidx, np.uint32), np.ascontiguousarray(b_vals, dt)]
        if len(arrs[0]) != K + 1 or len(arrs[3]) != K + 1:
            # reference: assert(csc.pos.size() == csr.pos.size()), SimSpGEMM.cpp:267
            raise OspError(_lib.ERR_DIM, f"pointer arrays must have K+1={K + 1} entries "
                                         f"(got {len(arrs[0])} and {len(arrs[3])})")
        cfg = self._config(validate, partial_capacity, k_range, row_shard)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_csc_csr(self._h, _DT[dt], M, K, N, *[_ptr(a) for a in arrs],
                                                 _lib.OSP_HOST, C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def spgemm_csc_csr_aos(self, M, K, N, a_pos, a_data, b_pos, b_data, *, validate=True, partial_capacity=0):
        """The reference's in-memory operands as they stand (``osp_spgemm_csc_csr_aos``): ``pos`` = ``CSRMatrix::pos``
        (K+1 uint64 offsets), ``data`` = ``CSRMatrix::data`` as a packed structured array ``aos_dtype(value dtype)`` =
        ``[('idx', '<u4'), ('val', '<f4' | '<f8')]`` (``common.h:10-16``: 8 or 12 bytes per record)."""
        vdt = np.dtype(a_data.dtype.fields["val"][0])
        want = aos_dtype(vdt)
        if a_data.dtype != want or b_data.dtype != want:
            raise TypeError(f"data arrays must both have dtype {want}")
        ap, bp = np.ascontiguousarray(a_pos, np.uint64), np.ascontiguousarray(b_pos, np.uint64)
        if len(ap) != K + 1 or len(bp) != K + 1:
            raise OspError(_lib.ERR_DIM, f"pos arrays must have K+1={K + 1} entries (got {len(ap)} and {len(bp)})")
        ad, bd = np.ascontiguousarray(a_data), np.ascontiguousarray(b_data)
        cfg = self._config(validate, partial_capacity, None)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_csc_csr_aos(self._h, _DT[vdt], M, K, N, _ptr(ap), _ptr(ad), _ptr(bp), _ptr(bd),
                                                     _lib.OSP_HOST, C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def alloc(self, nbytes):
        """Device memory from the context's buffer pool (``osp_context_alloc``); give it back with ``free``."""
        p = C.c_void_p()
        _lib.check(_lib.lib().osp_context_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value or 0

    def free(self, ptr):
        if self._h is not None and ptr:
            _lib.check(_lib.lib().osp_context_free(self._h, C.c_void_p(int(ptr))))

    def spgemm_coo(self, M, K, N, a, b, *, partial_capacity=0):
        """C = A * B from COO triples a = (rows, cols, vals), b = (rows, cols, vals) in any order (numpy, host):
        ``coo2csr<true>(A)`` / ``coo2csr(B)`` (SimSpGEMM.cpp:102-152) run on the GPU; duplicates raise 233."""
        dt = np.dtype(a[2].dtype)
        if dt not in _DT or np.dtype(b[2].dtype) != dt:
            raise TypeError("values must both be float32 or both float64")
        arrs = [np.ascontiguousarray(a[0], np.uint32), np.ascontiguousarray(a[1], np.uint32), np.ascontiguousarray(a[2], dt),
                np.ascontiguousarray(b[0], np.uint32), np.ascontiguousarray(b[1], np.uint32), np.ascontiguousarray(b[2], dt)]
        cfg = self._config(True, partial_capacity, None)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_coo(self._h, _DT[dt], M, K, N, len(arrs[0]), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]),
                                             len(arrs[3]), _ptr(arrs[3]), _ptr(arrs[4]), _ptr(arrs[5]), _lib.OSP_HOST,
                                             C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def stream_copy_gbps(self, nbytes=2 << 30, reps=10):
        """What a plain 16-bytes-per-lane copy reaches on this device, read + written bytes per second in GB/s
        (``osp_stream_copy_probe``): the measured roof beside the data sheet's."""
        g = C.c_double()
        _lib.check(_lib.lib().osp_stream_copy_probe(self._h, int(nbytes), int(reps), C.byref(g)))
        return g.value

    def spgemm_coo_device(self, dtype, M, K, N, nnz_a, a_ptrs, nnz_b, b_ptrs, *, partial_capacity=0, k_range=None):
        """``spgemm_coo`` on DEVICE arrays: a_ptrs / b_ptrs = (rows, cols, vals) addresses (u32, u32, dtype), entries in any
        order.  ``result.info['ms_ingest']`` is the device time of the two COO -> CSC / CSR conversions.  ``k_range``
        restricts the PRODUCT behind the conversions to a slab of k (the conversions always take the whole operands)."""
        cfg = self._config(True, partial_capacity, k_range)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_coo(self._h, _DT[np.dtype(dtype)], M, K, N, int(nnz_a), *[C.c_void_p(int(p)) for p in a_ptrs],
                                             int(nnz_b), *[C.c_void_p(int(p)) for p in b_ptrs], _lib.OSP_DEVICE, C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def spgemm_csc_csr_device(self, dtype, M, K, N, ptrs, *, validate=False, partial_capacity=0, k_range=None, row_shard=None):
        """Same with six DEVICE addresses (ints): a_colptr, a_rowidx, a_vals, b_rowptr, b_colidx, b_vals.
        The arrays must be COMPLETE when this is called: the context works on a stream of its own (or the one it was
        created on) and does not wait for kernels other streams -- e.g. torch's -- still have in flight on them."""
        cfg = self._config(validate, partial_capacity, k_range, row_shard)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_csc_csr(self._h, _DT[np.dtype(dtype)], M, K, N,
                                                 *[C.c_void_p(int(p)) for p in ptrs], _lib.OSP_DEVICE,
                                                 C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def spgemm_partials_device(self, dtype, M, K, N, ptrs, *, k_range=None):
        """The multiply phase alone (``osp_spgemm_partials``): the product's partial products, unmerged, as packed records
        grouped by output row.  ``ptrs`` = six DEVICE addresses as in ``spgemm_csc_csr_device``.  ``result.nnz`` = P;
        ``result.partials_ptrs()`` borrows the arrays."""
        cfg = self._config(False, 0, k_range)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_partials(self._h, _DT[np.dtype(dtype)], M, K, N, *[C.c_void_p(int(p)) for p in ptrs],
                                                  _lib.OSP_DEVICE, C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def merge_record_parts_device(self, dtype, M, N, part_ptrs, *, partial_capacity=0, validate=False):
        """Sum parts given as (rowptr, records) DEVICE addresses (``osp_merge_record_parts``) into one CSR.
        validate=True checks the offsets and the columns on the device first."""
        n = len(part_ptrs)
        rp, rc = (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, (r, c) in enumerate(part_ptrs):
            rp[i], rc[i] = int(r), int(c)
        cfg = self._config(validate, partial_capacity, None)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_merge_record_parts(self._h, _DT[np.dtype(dtype)], M, N, n, rp, rc, _lib.OSP_DEVICE, C.byref(cfg),
                                                     C.byref(h)))
        return CsrResult(self, h)

    def spgemm_csc_csr_panels(self, dtype, M, K, N, ptrs, on_panel, *, device=True, validate=False, partial_capacity=0,
                              k_range=None, row_shard=None):
        """Streamed product (``osp_spgemm_csc_csr_panels``): C is never resident as a whole.  ``ptrs`` = the six operand
        addresses (ints; device memory unless device=False).  ``on_panel(p)`` is called once per finished row panel, in
        row order, with a dict: row_begin, row_end, nnz, index, count and the DEVICE addresses rowptr / colidx / vals
        (valid only during the call; wrap them zero-copy, e.g. ``distributed._as_tensor``).  An exception raised by
        the callback aborts the product and is re-raised.  Returns the info dict (counters, phase times)."""
        cfg = self._config(validate, partial_capacity, k_range, row_shard)
        err = []

        def tramp(pp, _user):
            p = pp.contents
            try:
                on_panel({"row_begin": p.row_begin, "row_end": p.row_end, "nnz": p.nnz, "index": p.index, "count": p.count,
                          "rowptr": p.rowptr or 0, "colidx": p.colidx or 0, "vals": p.vals or 0})
                return 0
            except BaseException as e:  # never let an exception cross the C frames
                err.append(e)
                return 1

        fn = _lib.PANEL_FN(tramp)
        info = _lib.ResultInfo()
        st = _lib.lib().osp_spgemm_csc_csr_panels(self._h, _DT[np.dtype(dtype)], M, K, N, *[C.c_void_p(int(p)) for p in ptrs],
                                                  _lib.OSP_DEVICE if device else _lib.OSP_HOST, C.byref(cfg), fn, None, C.byref(info))
        if err:
            raise err[0]
        _lib.check(st)
        return info.as_dict()

    def merge_csr_parts(self, M, N, parts, *, partial_capacity=0):
        """Sum CSR matrices of equal shape.  parts: list of (rowptr, colidx, vals) numpy triples."""
        dt = np.dtype(parts[0][2].dtype)
        keep, n = [], len(parts)
        rp, ci, va = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, (r, c, v) in enumerate(parts):
            r = np.ascontiguousarray(r, np.int64); c = np.ascontiguousarray(c, np.uint32); v = np.ascontiguousarray(v, dt)
            keep += [r, c, v]
            rp[i], ci[i], va[i] = r.ctypes.data, (c.ctypes.data if c.size else 0), (v.ctypes.data if v.size else 0)
        cfg = self._config(False, partial_capacity, None)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_merge_csr_parts(self._h, _DT[dt], M, N, n, rp, ci, va, _lib.OSP_HOST,
                                                  C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def merge_csr_parts_device(self, dtype, M, N, part_ptrs, *, partial_capacity=0):
        """Device-resident parts: list of (rowptr, colidx, vals) device addresses."""
        n = len(part_ptrs)
        rp, ci, va = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, (r, c, v) in enumerate(part_ptrs):
            rp[i], ci[i], va[i] = int(r), int(c), int(v)
        cfg = self._config(False, partial_capacity, None)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_merge_csr_parts(self._h, _DT[np.dtype(dtype)], M, N, n, rp, ci, va,
                                                  _lib.OSP_DEVICE, C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)

    def spgemm_mtx(self, path_a, path_b, transpose_b=True, dtype=np.float32, *, validate=True, partial_capacity=0):
        """The reference CLI's data flow: two .mtx files in, A * B^T (default) out."""
        cfg = self._config(validate, partial_capacity, None)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_mtx(self._h, _DT[np.dtype(dtype)], os.fsencode(path_a), os.fsencode(path_b),
                                             int(bool(transpose_b)), C.byref(cfg), C.byref(h)))
        return CsrResult(self, h)


class MultiGpu:
    """Several GPUs of one node behind ONE call (``osp_multi_*``): the k-sharded product of SURVEY.md 8e inside the library --
    slabs of the shared dimension on their ranks, partial products copied GPU to GPU to the rank that owns their row while
    the next panel multiplies, one merge per row range as the pieces arrive.  ``devices`` may name an ordinal more than
    once (logical ranks sharing a GPU)."""

    def __init__(self, devices):
        self._h = None
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_multi_context_create(arr, len(devices), C.byref(h)))
        self._h = h
        self.devices = list(devices)
        self._ops = None

    def load(self, M, K, N, a_colptr, a_rowidx, a_vals, b_rowptr, b_colidx, b_vals):
        """Cut k into slabs and put every rank's slab on its GPU (host numpy operands).  Replaces what was loaded before."""
        dt = np.dtype(a_vals.dtype)
        if dt not in _DT or np.dtype(b_vals.dtype) != dt:
            raise TypeError("values must both be float32 or both float64")
        arrs = [np.ascontiguousarray(a_colptr, np.int64), np.ascontiguousarray(a_rowidx, np.uint32), np.ascontiguousarray(a_vals, dt),
                np.ascontiguousarray(b_rowptr, np.int64), np.ascontiguousarray(b_colidx, np.uint32), np.ascontiguousarray(b_vals, dt)]
        if len(arrs[0]) != K + 1 or len(arrs[3]) != K + 1:
            raise OspError(_lib.ERR_DIM, f"pointer arrays must have K+1={K + 1} entries (got {len(arrs[0])} and {len(arrs[3])})")
        self.unload()
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_multi_operands_create(self._h, _DT[dt], M, K, N, *[_ptr(a) for a in arrs], C.byref(h)))
        self._ops, self._shape, self._dtype = h, (M, N), dt

    def unload(self):
        if self._ops is not None:
            _lib.lib().osp_multi_operands_destroy(self._ops)
            self._ops = None

    def multiply(self, *, validate=False, partial_capacity=0, fetch=True, checksum=False):
        """One product of the loaded operands.  Returns (info dict, (rowptr, colidx, vals) host arrays or None).
        checksum=True adds info["val_sum"] = the sum of all values of C, formed shard by shard on the GPUs (torch)."""
        cfg = _lib.Config()
        _lib.lib().osp_config_default(C.byref(cfg))
        cfg.validate = int(bool(validate))
        cfg.partial_capacity = int(partial_capacity or 0)
        h = C.c_void_p()
        _lib.check(_lib.lib().osp_spgemm_multi(self._h, self._ops, C.byref(cfg), C.byref(h)))
        try:
            info = _lib.MultiInfo()
            _lib.check(_lib.lib().osp_multi_result_info(h, C.byref(info)))
            out = None
            extra = {}
            if checksum:
                import torch
                from .distributed import _as_tensor
                total = 0.0
                for g in range(info.nranks):
                    sh, va = C.c_void_p(), C.c_void_p()
                    _lib.check(_lib.lib().osp_multi_result_shard(h, g, None, None, C.byref(sh)))
                    _lib.check(_lib.lib().osp_result_device_ptrs(sh, None, None, C.byref(va)))
                    n = info.rank[g].nnz_c
                    if n:
                        dev = torch.device("cuda", info.rank[g].device)
                        f64 = self._dtype == np.float64
                        total += float(_as_tensor(va.value, n, "<f8" if f64 else "<f4", dev, torch.float64 if f64 else torch.float32)
                                       .sum(dtype=torch.float64))
                extra["val_sum"] = total
            if fetch:
                rowptr = np.zeros(self._shape[0] + 1, np.int64)
                colidx = np.empty(info.nnz_c, np.uint32)
                vals = np.empty(info.nnz_c, self._dtype)
                _lib.check(_lib.lib().osp_multi_result_copy_csr(h, _ptr(rowptr), _ptr(colidx), _ptr(vals)))
                out = (rowptr, colidx, vals)
            d = info.as_dict()
            d.update(extra)
            return d, out
        finally:
            _lib.lib().osp_multi_result_destroy(h)

    def spgemm_csc_csr(self, M, K, N, a_colptr, a_rowidx, a_vals, b_rowptr, b_colidx, b_vals, **kw):
        self.load(M, K, N, a_colptr, a_rowidx, a_vals, b_rowptr, b_colidx, b_vals)
        return self.multiply(**kw)

    def close(self):
        if self._h is not None:
            self.unload()
            _lib.lib().osp_multi_context_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- ingest helpers (host, no GPU) ---------------------------------------------------------------
def read_mtx(path, symmetric=False):
    """``readcoo`` (SimSpGEMM.cpp:55-100): returns (nrow, ncol, rows u32, cols u32, vals f64)."""
    L = _lib.lib()
    nrow, ncol, nnz = C.c_uint64(), C.c_uint64(), C.c_uint64()
    r, c, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(L.osp_mtx_read(os.fsencode(path), int(symmetric), C.byref(nrow), C.byref(ncol), C.byref(nnz),
                              C.byref(r), C.byref(c), C.byref(v)))
    n = nnz.value
    try:
        rows = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_uint32)), (max(n, 1),))[:n].copy()
        cols = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint32)), (max(n, 1),))[:n].copy()
        vals = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), (max(n, 1),))[:n].copy()
    finally:
        for p in (r, c, v):
            L.osp_host_free(p)
    return nrow.value, ncol.value, rows, cols, vals


def _compress(by_col, nseg, rows, cols, vals):
    dt = np.dtype(vals.dtype)
    if dt not in _DT:
        raise TypeError("values must be float32 or float64")
    rows = np.ascontiguousarray(rows, np.uint32); cols = np.ascontiguousarray(cols, np.uint32)
    vals = np.ascontiguousarray(vals)
    nnz = len(rows)
    ptr = np.zeros(nseg + 1, np.int64); idx = np.zeros(nnz, np.uint32); out = np.zeros(nnz, dt)
    fn = getattr(_lib.lib(), "osp_coo_to_compressed_f32" if dt == np.float32 else "osp_coo_to_compressed_f64")
    _lib.check(fn(int(by_col), nseg, nnz, _ptr(rows), _ptr(cols), _ptr(vals), _ptr(ptr), _ptr(idx), _ptr(out)))
    return ptr, idx, out


def coo_to_csr(nrow, rows, cols, vals):
    """``coo2csr<false>`` (SimSpGEMM.cpp:102-152); duplicate coordinates raise OspError(233)."""
    return _compress(0, nrow, rows, cols, vals)


def coo_to_csc(ncol, rows, cols, vals):
    """``coo2csr<true>`` (SimSpGEMM.cpp:102-152)."""
    return _compress(1, ncol, rows, cols, vals)


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _as_coo(x):
    """dense ndarray / torch tensor / scipy sparse -> (nrow, ncol, rows, cols, vals)."""
    import scipy.sparse as sp
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    if not sp.issparse(x):
        x = sp.csr_matrix(np.asarray(x))  # what util.py:61-62 does before mmwrite
    x = x.tocoo()
    return x.shape[0], x.shape[1], x.row.astype(np.uint32), x.col.astype(np.uint32), x.data


def spgemm(A, B, transpose_b=True, ctx=None, dtype=None):
    """``A @ B.T`` (default, as the reference CLI) or ``A @ B`` as scipy CSR, computed on the GPU."""
    ctx = ctx or default_context()
    M, K, ar, ac, av = _as_coo(A)
    br_n, bc_n, br, bc, bv = _as_coo(B)
    if transpose_b:
        br_n, bc_n, br, bc = bc_n, br_n, bc, br
    if br_n != K:
        raise OspError(_lib.ERR_DIM, f"inner dimensions differ: A is {M}x{K}, B is {br_n}x{bc_n}")
    dt = np.dtype(dtype or np.result_type(av.dtype, bv.dtype))
    if dt not in _DT:
        dt = np.dtype(np.float64)
    a = coo_to_csc(K, ar, ac, av.astype(dt))
    b = coo_to_csr(K, br, bc, bv.astype(dt))
    with_res = ctx.spgemm_csc_csr(M, K, bc_n, *a, *b)
    out = with_res.to_scipy()
    with_res.close()
    return out
