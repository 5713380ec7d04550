"""CPU-side checks of the boundary: the library loads, exports every symbol the header declares, the
host-only entry points (reader, COO->compressed) match the reference's goldens, and with no GPU the
compute entry points fail loudly instead of falling back."""
import os
import re

import numpy as np
import pytest

from outerspace_amd import _lib
from outerspace_amd import spgemm as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "outerspace_spgemm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only, not prose in comments
    declared = set(re.findall(r"\b(osp_[a-z0-9_]+)\s*\(", hdr))
    L = _lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)


def test_struct_layouts_match_header():
    import ctypes
    # osp_config_t: int, u64, u64, u64, int[8]; osp_result_info_t ends with two u32 and an int
    assert ctypes.sizeof(_lib.Config) == 8 + 8 + 8 + 8 + 32
    assert ctypes.sizeof(_lib.Panel) == 3 * 8 + 3 * 8 + 4 * 4  # osp_panel_t: three u64, three pointers, four u32
    cfg = _lib.Config()
    _lib.lib().osp_config_default(ctypes.byref(cfg))
    assert cfg.validate == 1 and cfg.partial_capacity == 0 and cfg.k_end == 0
    assert _lib.lib().osp_status_string(233).decode().startswith("duplicate")


def test_ctypes_structs_have_the_sizes_the_c_compiler_gives(tmp_path):
    """Every struct the Python binding mirrors, measured by the C compiler on the header itself (an appended field that the
    binding forgets would shift everything behind it)."""
    import ctypes
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "outerspace_spgemm.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(osp_config_t), sizeof(osp_result_info_t), sizeof(osp_panel_t), '
                   'sizeof(osp_multi_rank_info_t), sizeof(osp_multi_info_t), offsetof(osp_result_info_t, rank_atomic), '
                   'offsetof(osp_multi_info_t, rank)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(_lib.Config), ctypes.sizeof(_lib.ResultInfo), ctypes.sizeof(_lib.Panel), ctypes.sizeof(_lib.MultiRankInfo),
            ctypes.sizeof(_lib.MultiInfo), _lib.ResultInfo.rank_atomic.offset, _lib.MultiInfo.rank.offset]
    assert got == want, (got, want)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(S.OspError) as ei:
        S.Context(0)
    assert ei.value.status == _lib.ERR_HIP and "no CPU path" in str(ei.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "outerspace_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                # comments may mention the oracle; code must never import, load or link it
                assert not re.search(r"import\s+oracle|from\s+oracle|liboracle|oracle[/.]|_ref/|libref", text), (dirpath, f)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_reader_matches_reference_golden(golden_dir, dt):
    g = np.load(os.path.join(golden_dir, "reader_quirks_expected.npz"))
    nrow, ncol, r, c, v = S.read_mtx(os.path.join(golden_dir, "reader_quirks.mtx"))
    s = np.dtype(dt).name
    assert (nrow, ncol) == (int(g["nrow"]), int(g["ncol"]))
    assert np.array_equal(r, g[f"rows_{s}"]) and np.array_equal(c, g[f"cols_{s}"])
    assert np.array_equal(v.astype(dt), g[f"vals_{s}"])


def test_reader_c1_and_mlp_files(golden_dir, port):
    for name in ("c1_A.mtx", "mlp_fc1_weight.mtx"):
        a = S.read_mtx(os.path.join(golden_dir, name))
        b = port.readcoo(os.path.join(golden_dir, name))
        assert a[:2] == b[:2]
        for x, y in zip(a[2:], b[2:]):
            assert np.array_equal(x, y)
    with pytest.raises(S.OspError) as ei:
        S.read_mtx(os.path.join(golden_dir, "does_not_exist.mtx"))
    assert ei.value.status == _lib.ERR_IO


def test_host_conversion_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "edges_expected.npz"))
    for tr, nseg, fn in ((0, 6, S.coo_to_csr), (1, 8, S.coo_to_csc)):
        ptr, idx, val = fn(nseg, g["conv_rows"], g["conv_cols"], g["conv_vals"])
        assert np.array_equal(ptr, g[f"conv{tr}_pos"]) and np.array_equal(idx, g[f"conv{tr}_idx"])
        assert np.array_equal(val, g[f"conv{tr}_val"])
    for fn in (S.coo_to_csr, S.coo_to_csc):
        with pytest.raises(S.OspError) as ei:
            fn(3, g["dup_rows"], g["dup_cols"], g["dup_vals"])
        assert ei.value.status == 233
    # documented divergence from the reference's back-fill quirk (SimSpGEMM.cpp:143-148)
    ptr, idx, _ = S.coo_to_csr(4, g["onerow_rows"], g["onerow_cols"], g["onerow_vals"])
    assert np.array_equal(ptr, [0, 0, 0, 3, 3]) and np.array_equal(idx, [0, 1, 3])
    with pytest.raises(S.OspError) as ei:
        S.coo_to_csr(2, np.array([5], np.uint32), np.array([0], np.uint32), np.array([1.0]))
    assert ei.value.status == _lib.ERR_RANGE
    # rect case: conversion equals the reference's coo2csr for both orientations and value types
    for dt in (np.float32, np.float64):
        s = np.dtype(dt).name
        a = S.coo_to_csc(7, g["rect_a_rows"], g["rect_a_cols"], g["rect_a_vals"].astype(dt))
        b = S.coo_to_csr(7, g["rect_b_rows"], g["rect_b_cols"], g["rect_b_vals"].astype(dt))
        for got, key in zip((*a, *b), ("apos", "aidx", "aval", "bpos", "bidx", "bval")):
            assert np.array_equal(got, g[f"rect_{key}_{s}"]), key


def test_host_layer_under_sanitizers(golden_dir, tmp_path):
    """the host C++ layer (reader, COO -> CSC/CSR, the CLI's host parts) and the plain-C oracle under
    AddressSanitizer + UBSan.  `make asan` builds tests/asan_host_driver.cpp with osp_host.cpp and oracle_spgemm.c;
    the driver feeds them the golden files and hostile inputs (huge header counts, garbage, megabyte lines) and compares
    product and oracle; a sanitizer report aborts it.  CPU only -- GPU sanitizers are not available on this pool."""
    import subprocess
    csrc = os.path.join(ROOT, "outerspace_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asan"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "outerspace_amd", "osp_host_asan_test"), golden_dir, str(tmp_path)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_cli_under_sanitizers_fails_cleanly_without_gpu(golden_dir, tmp_path):
    """The CLI built with ASan/UBSan against the real library: usage error, unreadable file, and -- where there is no GPU --
    a clean "no CPU path" failure at context creation instead of a fallback."""
    import subprocess
    import torch
    csrc = os.path.join(ROOT, "outerspace_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asan"], check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "outerspace_amd", "osp_spgemm_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")  # (the HIP runtime keeps process-lifetime allocations)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, str(tmp_path / "nope.mtx"), str(tmp_path / "nope.mtx")], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 1 and "cannot open" in r.stderr and "AddressSanitizer" not in r.stderr
    if not torch.cuda.is_available():
        a = os.path.join(golden_dir, "c1_A.mtx")
        r = subprocess.run([exe, a, os.path.join(golden_dir, "c1_B.mtx")], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 1 and "no CPU path" in r.stderr and "NNZ = 410" in r.stdout
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_integration_doc_quotes_the_compiled_binding():
    """INTEGRATION.md section 2 shows integration/simspgemm_gpu_binding.h's function verbatim (the header is compiled
    against the reference's common.h by `make -C oracle`, so the documented binding cannot drift from a working one)."""
    hdr = open(os.path.join(ROOT, "integration", "simspgemm_gpu_binding.h")).read()
    fn = hdr[hdr.index("inline COOMatrix cscMulcsrMergedGPU"):]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert fn in doc
    if os.path.exists("/root/reference/simulator/common.h"):   # build container: the demo that uses it has been built
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
        for sfx in ("f32", "f64"):
            assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", f"binding_demo_{sfx}"))


@pytest.mark.parametrize("threads", ["1", "3", "8"])
def test_parallel_reader_gives_the_single_threaded_result(tmp_path, monkeypatch, port, threads):
    """osp_mtx_read parses the lines behind the header on several threads (byte ranges cut at line starts) and concatenates
    the pieces in file order: whatever the thread count, the result is what the oracle's one-pass readcoo gives -- comment and
    blank lines in the middle, pattern entries, CRLF ends, no newline at the end of the file, fewer lines than threads."""
    rng = np.random.default_rng(12)
    lines = ["%%MatrixMarket matrix coordinate real general", "% a comment", "", "300 200 9999"]
    for i in range(5000):
        r, c = int(rng.integers(1, 301)), int(rng.integers(1, 201))
        kind = i % 11
        if kind == 0:
            lines.append(f"{r} {c}")                      # pattern entry: value 1.0 (SimSpGEMM.cpp:92-93)
        elif kind == 1:
            lines.append(f"  {r}\t{c}   {rng.uniform(-3, 3):.17g}\r")   # blanks, a tab, CRLF
        else:
            lines.append(f"{r} {c} {rng.uniform(-3, 3):.9g}")
        if i % 97 == 0:
            lines.append("% comment in the middle")
        if i % 131 == 0:
            lines.append("   ")
    big = tmp_path / "big.mtx"
    big.write_text("\n".join(lines))                      # no newline at the end
    small = tmp_path / "small.mtx"
    small.write_text("%%MatrixMarket matrix coordinate real general\n4 4 2\n1 2 0.5\n4 4 -1e3")
    monkeypatch.setenv("OSP_PARSE_THREADS", threads)
    for path in (big, small):
        got = S.read_mtx(str(path))
        want = port.readcoo(str(path))
        assert got[:2] == want[:2]
        for x, y in zip(got[2:], want[2:]):
            assert np.array_equal(x, y)
