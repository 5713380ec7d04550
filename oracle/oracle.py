"""ctypes bindings for the CPU checker -- TEST INFRASTRUCTURE ONLY.

Two libraries, both built by ``make -C oracle``:

* ``liboracle_spgemm.so``  -- the plain-C restatement (``oracle_spgemm.c``), kind "port".
* ``_ref/libref_f{32,64}.so`` -- the reference's own ``simulator/SimSpGEMM.cpp`` compiled in
  place (``ref_driver.cpp``), kind "reference".  Present only where it was built from
  ``/root/reference``; the prebuilt files travel to the GPU box with the snapshot.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``outerspace_amd/`` does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ERR_DUPLICATE = 233

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build(ref=True):
    """(Re)build the oracle libraries.  Building the checker is not using it."""
    subprocess.run(["make", "-C", _HERE] + ([] if ref else ["liboracle_spgemm.so"]),
                   check=True, stdout=subprocess.DEVNULL)


def _vp(dtype):
    return np.ctypeslib.ndpointer(dtype, flags="C_CONTIGUOUS")


def _dt(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return dtype, "f32", C.c_float
    if dtype == np.float64:
        return dtype, "f64", C.c_double
    raise TypeError(f"value dtype must be float32/float64, got {dtype}")


class _Port:
    """liboracle_spgemm.so"""

    def __init__(self):
        path = os.path.join(_HERE, "liboracle_spgemm.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.osp_oracle_free.argtypes = [C.c_void_p]
        self.lib.osp_oracle_mulflops_f64.restype = C.c_uint64
        self.lib.osp_oracle_mulflops_f32.restype = C.c_uint64

    def readcoo(self, path, sym=False):
        nrow, ncol, nnz = C.c_uint64(), C.c_uint64(), C.c_uint64()
        r, c, v = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_double)()
        rc = self.lib.osp_oracle_readcoo(os.fsencode(path), int(sym), C.byref(nrow), C.byref(ncol),
                                         C.byref(nnz), C.byref(r), C.byref(c), C.byref(v))
        if rc:
            raise OSError(f"osp_oracle_readcoo({path}) -> {rc}")
        n = nnz.value
        rows = np.ctypeslib.as_array(r, (max(n, 1),))[:n].copy()
        cols = np.ctypeslib.as_array(c, (max(n, 1),))[:n].copy()
        vals = np.ctypeslib.as_array(v, (max(n, 1),))[:n].copy()
        for p in (r, c, v):
            self.lib.osp_oracle_free(p)
        return nrow.value, ncol.value, rows, cols, vals

    def coo2csr(self, transpose, nseg, rows, cols, vals):
        dtype, sfx, _ = _dt(vals.dtype)
        rows = np.ascontiguousarray(rows, np.uint32)
        cols = np.ascontiguousarray(cols, np.uint32)
        vals = np.ascontiguousarray(vals)
        nnz = len(rows)
        pos = np.zeros(nseg + 1, np.int64)
        idx = np.zeros(max(nnz, 1), np.uint32)
        out = np.zeros(max(nnz, 1), dtype)
        fn = getattr(self.lib, f"osp_oracle_coo2csr_{sfx}")
        fn.argtypes = [C.c_int, C.c_size_t, C.c_size_t, _u32p, _u32p, _vp(dtype), _i64p, _u32p,
                       _vp(dtype)]
        rc = fn(int(transpose), nseg, nnz, rows if nnz else np.zeros(1, np.uint32),
                cols if nnz else np.zeros(1, np.uint32), vals if nnz else np.zeros(1, dtype),
                pos, idx, out)
        return rc, pos, idx[:nnz], out[:nnz]

    def spgemm(self, M, K, N, a_colptr, a_rowidx, a_val, b_rowptr, b_colidx, b_val, k0=0, k1=None):
        """Returns dict(rowptr, colidx, vals, partials, secs)."""
        dtype, sfx, cty = _dt(a_val.dtype)
        k1 = K if k1 is None else k1
        a_colptr = np.ascontiguousarray(a_colptr, np.int64)
        b_rowptr = np.ascontiguousarray(b_rowptr, np.int64)
        pad = lambda x, t: np.ascontiguousarray(x, t) if len(x) else np.zeros(1, t)
        rowptr = np.zeros(M + 1, np.int64)
        cc, cv = C.POINTER(C.c_uint32)(), C.POINTER(cty)()
        P = C.c_uint64()
        secs = (C.c_double * 2)()
        fn = getattr(self.lib, f"osp_oracle_spgemm_{sfx}")
        fn.argtypes = [C.c_size_t] * 5 + [_i64p, _u32p, _vp(dtype), _i64p, _u32p, _vp(dtype), _i64p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = fn(M, K, N, k0, k1, a_colptr, pad(a_rowidx, np.uint32), pad(a_val, dtype), b_rowptr,
                pad(b_colidx, np.uint32), pad(b_val, dtype), rowptr, C.byref(cc), C.byref(cv),
                C.byref(P), secs)
        if rc:
            raise RuntimeError(f"osp_oracle_spgemm_{sfx} -> {rc}")
        n = int(rowptr[M])
        colidx = np.ctypeslib.as_array(cc, (max(n, 1),))[:n].copy()
        vals = np.ctypeslib.as_array(cv, (max(n, 1),))[:n].copy()
        self.lib.osp_oracle_free(cc)
        self.lib.osp_oracle_free(cv)
        return dict(rowptr=rowptr, colidx=colidx, vals=vals, partials=P.value,
                    secs=(secs[0], secs[1]))

    def mulflops(self, K, a_colptr, b_rowptr, k0=0, k1=None):
        k1 = K if k1 is None else k1
        fn = self.lib.osp_oracle_mulflops_f64
        fn.argtypes = [C.c_size_t, C.c_size_t, _i64p, _i64p]
        return fn(k0, k1, np.ascontiguousarray(a_colptr, np.int64),
                  np.ascontiguousarray(b_rowptr, np.int64))


class _Ref:
    """oracle/_ref/libref_f{32,64}.so -- the compiled reference."""

    def __init__(self, dtype):
        self.dtype, self.sfx, self.cty = _dt(dtype)
        path = os.path.join(_HERE, "_ref", f"libref_{self.sfx}.so")
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} missing: run `make -C oracle` where /root/reference is present")
        self.lib = C.CDLL(path)
        assert self.lib.osp_ref_value_size() == self.dtype.itemsize
        self.lib.osp_ref_free.argtypes = [C.c_void_p]

    def _take(self, n, r, c, v):
        rows = np.ctypeslib.as_array(r, (max(n, 1),))[:n].copy()
        cols = np.ctypeslib.as_array(c, (max(n, 1),))[:n].copy()
        vals = np.ctypeslib.as_array(v, (max(n, 1),))[:n].copy()
        for p in (r, c, v):
            self.lib.osp_ref_free(p)
        return rows, cols, vals

    def readcoo(self, path, sym=False):
        nrow, ncol, nnz = C.c_uint64(), C.c_uint64(), C.c_uint64()
        r, c, v = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(self.cty)()
        rc = self.lib.osp_ref_readcoo(os.fsencode(path), int(sym), C.byref(nrow), C.byref(ncol),
                                      C.byref(nnz), C.byref(r), C.byref(c), C.byref(v))
        if rc:
            raise OSError(f"osp_ref_readcoo({path}) -> {rc}")
        return (nrow.value, ncol.value) + self._take(nnz.value, r, c, v)

    def coo2csr(self, transpose, nseg, rows, cols, vals):
        nnz = len(rows)
        pos = np.zeros(nseg + 1, np.int64)
        idx = np.zeros(max(nnz, 1), np.uint32)
        out = np.zeros(max(nnz, 1), self.dtype)
        fn = self.lib.osp_ref_coo2csr
        fn.argtypes = [C.c_int, C.c_uint64, C.c_uint64, _u32p, _u32p, _vp(self.dtype), _i64p, _u32p,
                       _vp(self.dtype)]
        pad = lambda x, t: np.ascontiguousarray(x, t) if len(x) else np.zeros(1, t)
        rc = fn(int(transpose), nseg, nnz, pad(rows, np.uint32), pad(cols, np.uint32),
                pad(vals, self.dtype), pos, idx, out)
        return rc, pos, idx[:nnz], out[:nnz]

    def spgemm_mtx(self, path_a, path_b, transpose_b=True):
        """(rc, M, N, partials, rows, cols, vals) -- COO sorted by (row, col)."""
        M, N, nnzc, P = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        r, c, v = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(self.cty)()
        rc = self.lib.osp_ref_spgemm_mtx(os.fsencode(path_a), os.fsencode(path_b), int(transpose_b),
                                         C.byref(M), C.byref(N), C.byref(nnzc), C.byref(P),
                                         C.byref(r), C.byref(c), C.byref(v))
        if rc:
            return rc, 0, 0, 0, None, None, None
        return (0, M.value, N.value, P.value) + self._take(nnzc.value, r, c, v)

    def spgemm_csx(self, K, a_colptr, a_rowidx, a_val, b_rowptr, b_colidx, b_val, k0=0, k1=None,
                   timing_only=False):
        """cscMulcsr + sort/sum on compressed operands.  dict(rows, cols, vals, nnzc, partials, secs)."""
        k1 = K if k1 is None else k1
        pad = lambda x, t: np.ascontiguousarray(x, t) if len(x) else np.zeros(1, t)
        nnzc, P = C.c_uint64(), C.c_uint64()
        r, c, v = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(self.cty)()
        secs = (C.c_double * 2)()
        fn = self.lib.osp_ref_spgemm_csx
        fn.argtypes = [C.c_uint64] * 3 + [_i64p, _u32p, _vp(self.dtype), _i64p, _u32p,
                                          _vp(self.dtype)] + [C.c_void_p] * 6
        rc = fn(K, k0, k1, np.ascontiguousarray(a_colptr, np.int64), pad(a_rowidx, np.uint32),
                pad(a_val, self.dtype), np.ascontiguousarray(b_rowptr, np.int64),
                pad(b_colidx, np.uint32), pad(b_val, self.dtype), C.byref(nnzc), C.byref(P),
                None if timing_only else C.byref(r), None if timing_only else C.byref(c),
                None if timing_only else C.byref(v), secs)
        if rc:
            raise RuntimeError(f"osp_ref_spgemm_csx -> {rc}")
        out = dict(nnzc=nnzc.value, partials=P.value, secs=(secs[0], secs[1]))
        if not timing_only:
            out["rows"], out["cols"], out["vals"] = self._take(nnzc.value, r, c, v)
        return out


    #: osp_ref_spgemm_variant: which of the reference's producers / mergers forms the product
    VARIANTS = {
        "cscMulcsr+deduplicateCOO": 0,            # SimSpGEMM.cpp:265-281 + :519-535 (the path the build replaces)
        "csc2rawcompact+compactMulcsr": 1,        # :221-242 + :247-263, merged by deduplicateCOO
        "csr2compact+compactMulcsr": 2,           # :154-219 + :247-263, merged by deduplicateCOO
        "csr2compact+merge": 3,                   # :154-219 + merge/mergeHardware/multHardware/merge2way :306-517
    }

    def spgemm_variant(self, variant, M, K, a_colptr, a_rowidx, a_val, b_rowptr, b_colidx, b_val):
        """The product through one of the reference's own alternative formulations (``VARIANTS``).
        Returns (rc, dict(rows, cols, vals, nnzc, partials)); rc 6 = the reference's merge() would trip its
        own assert for this input (see ref_driver.cpp), rc 233 = its dupcheck threw."""
        pad = lambda x, t: np.ascontiguousarray(x, t) if len(x) else np.zeros(1, t)
        nnzc, P = C.c_uint64(), C.c_uint64()
        r, c, v = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(self.cty)()
        fn = self.lib.osp_ref_spgemm_variant
        fn.argtypes = [C.c_int, C.c_uint64, C.c_uint64, _i64p, _u32p, _vp(self.dtype), _i64p, _u32p,
                       _vp(self.dtype)] + [C.c_void_p] * 5
        rc = fn(int(self.VARIANTS.get(variant, variant)), M, K, np.ascontiguousarray(a_colptr, np.int64),
                pad(a_rowidx, np.uint32), pad(a_val, self.dtype), np.ascontiguousarray(b_rowptr, np.int64),
                pad(b_colidx, np.uint32), pad(b_val, self.dtype), C.byref(nnzc), C.byref(P), C.byref(r), C.byref(c),
                C.byref(v))
        if rc:
            return rc, None
        out = dict(nnzc=nnzc.value, partials=P.value)
        out["rows"], out["cols"], out["vals"] = self._take(nnzc.value, r, c, v)
        return 0, out


_port = None
_refs = {}


def port():
    global _port
    if _port is None:
        _port = _Port()
    return _port


def ref(dtype):
    key = np.dtype(dtype).name
    if key not in _refs:
        _refs[key] = _Ref(dtype)
    return _refs[key]


def have_ref():
    return all(os.path.exists(os.path.join(_HERE, "_ref", f"libref_{s}.so")) for s in ("f32", "f64"))


def coo_to_csr(M, rows, cols, vals):
    """Sorted COO -> (rowptr, colidx, vals)."""
    rowptr = np.zeros(M + 1, np.int64)
    np.add.at(rowptr, np.asarray(rows, np.int64) + 1, 1)
    return np.cumsum(rowptr), np.asarray(cols, np.uint32), vals
