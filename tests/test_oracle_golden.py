"""Pins oracle/oracle_spgemm.c (the CPU restatement) against golden vectors produced by the
reference's own compiled functions (tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from outerspace_amd import generators as gen
from oracle import oracle as orc


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def spgemm_from_coo(port, M, K, N, a, b, dt):
    rc, apos, aidx, aval = port.coo2csr(True, K, a[0], a[1], a[2].astype(dt))
    assert rc == 0
    rc, bpos, bidx, bval = port.coo2csr(False, K, b[0], b[1], b[2].astype(dt))
    assert rc == 0
    return port.spgemm(M, K, N, apos, aidx, aval, bpos, bidx, bval), (apos, aidx, aval, bpos, bidx, bval)


def csr_rows(rowptr):
    return np.repeat(np.arange(len(rowptr) - 1, dtype=np.uint32), np.diff(rowptr))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_reader_matches_reference(port, golden_dir, dt):
    g = load(golden_dir, "reader_quirks_expected.npz")
    nrow, ncol, r, c, v = port.readcoo(os.path.join(golden_dir, "reader_quirks.mtx"))
    s = np.dtype(dt).name
    assert (nrow, ncol) == (int(g["nrow"]), int(g["ncol"]))
    assert np.array_equal(r, g[f"rows_{s}"]) and np.array_equal(c, g[f"cols_{s}"])
    # the reference narrows the parsed double with value_t(val), SimSpGEMM.cpp:94
    assert np.array_equal(v.astype(dt), g[f"vals_{s}"])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("transpose_b", [True, False])
def test_c1_exact(port, golden_dir, dt, transpose_b):
    """BASELINE configs[0]: 64x64 10 % dense through .mtx, A*B^T (reference CLI) and A*B."""
    if not transpose_b and dt == np.float32:
        pytest.skip("A*B golden stored for f64 only")
    g = load(golden_dir, "c1_expected.npz")
    _, _, ar, ac, av = port.readcoo(os.path.join(golden_dir, "c1_A.mtx"))
    _, _, br, bc, bv = port.readcoo(os.path.join(golden_dir, "c1_B.mtx"))
    if transpose_b:
        br, bc = bc, br
    res, _ = spgemm_from_coo(port, 64, 64, 64, (ar, ac, av), (br, bc, bv), dt)
    s = np.dtype(dt).name
    pre = "" if transpose_b else "nt_"
    rows = g[f"rows_{s}"] if transpose_b else g["nt_rows"]
    cols = g[f"cols_{s}"] if transpose_b else g["nt_cols"]
    vals = g[f"vals_{s}"] if transpose_b else g["nt_vals"]
    assert res["partials"] == int(g["P"] if transpose_b else g[pre + "P"])
    assert np.array_equal(csr_rows(res["rowptr"]), rows)
    assert np.array_equal(res["colidx"], cols)
    tol = 1e-5 if dt == np.float32 else 1e-12
    assert np.allclose(res["vals"], vals, rtol=tol, atol=0)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_rect_with_empty_segments(port, golden_dir, dt):
    g = load(golden_dir, "edges_expected.npz")
    s = np.dtype(dt).name
    a = (g["rect_a_rows"], g["rect_a_cols"], g["rect_a_vals"])
    b = (g["rect_b_rows"], g["rect_b_cols"], g["rect_b_vals"])
    res, csx = spgemm_from_coo(port, 5, 7, 3, a, b, dt)
    # conversion equals the reference's coo2csr
    for got, key in zip(csx, ("apos", "aidx", "aval", "bpos", "bidx", "bval")):
        assert np.array_equal(got, g[f"rect_{key}_{s}"]), key
    assert res["partials"] == int(g[f"rect_partials_{s}"])
    assert np.array_equal(csr_rows(res["rowptr"]), g[f"rect_rows_{s}"])
    assert np.array_equal(res["colidx"], g[f"rect_cols_{s}"])
    assert np.allclose(res["vals"], g[f"rect_vals_{s}"], rtol=1e-6, atol=0)


def test_cancellation_keeps_explicit_zero(port, golden_dir):
    g = load(golden_dir, "edges_expected.npz")
    a = (g["cancel_a_rows"], g["cancel_a_cols"], g["cancel_a_vals"])
    b = (g["cancel_b_rows"], g["cancel_b_cols"], g["cancel_b_vals"])
    res, _ = spgemm_from_coo(port, 2, 2, 2, a, b, np.float64)
    assert np.array_equal(csr_rows(res["rowptr"]), g["cancel_rows"])
    assert np.array_equal(res["colidx"], g["cancel_cols"])
    assert np.array_equal(res["vals"], g["cancel_vals"])
    assert 0.0 in res["vals"]


def test_duplicate_is_233(port, golden_dir):
    g = load(golden_dir, "edges_expected.npz")
    assert int(g["dup_rc_csr"]) == 233 and int(g["dup_rc_csc"]) == 233
    for tr in (False, True):
        rc, *_ = port.coo2csr(tr, 3, g["dup_rows"], g["dup_cols"], g["dup_vals"])
        assert rc == orc.ERR_DUPLICATE


def test_conversion_unsorted_trailing_empty(port, golden_dir):
    g = load(golden_dir, "edges_expected.npz")
    for tr, nseg in ((0, 6), (1, 8)):
        rc, pos, idx, val = port.coo2csr(bool(tr), nseg, g["conv_rows"], g["conv_cols"], g["conv_vals"])
        assert rc == 0 == int(g[f"conv{tr}_rc"])
        assert np.array_equal(pos, g[f"conv{tr}_pos"])
        assert np.array_equal(idx, g[f"conv{tr}_idx"])
        assert np.array_equal(val, g[f"conv{tr}_val"])


def test_one_segment_quirk_is_a_documented_divergence(port, golden_dir):
    """Reference back-fill (SimSpGEMM.cpp:143-148) empties a matrix with ONE non-empty row."""
    g = load(golden_dir, "edges_expected.npz")
    assert np.array_equal(g["onerow_ref_pos"], [3, 3, 3, 3, 3])       # what the reference does
    rc, pos, idx, val = port.coo2csr(False, 4, g["onerow_rows"], g["onerow_cols"], g["onerow_vals"])
    assert rc == 0 and np.array_equal(pos, [0, 0, 0, 3, 3])           # what we do (exact)
    assert np.array_equal(idx, [0, 1, 3])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_rmat10_digest(port, golden_dir, dt):
    g = load(golden_dir, "rmat10_expected.npz")
    s = np.dtype(dt).name
    n, rows, cols, vals = gen.rmat_coo(10, 16, "g500", seed=1, dtype=dt)
    assert sha(rows) + sha(cols) + sha(vals) == str(g[f"in_sha_{s}"]), "generator drifted"
    res, _ = spgemm_from_coo(port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt)
    assert res["partials"] == int(g[f"P_{s}"])
    assert len(res["colidx"]) == int(g[f"nnzC_{s}"])
    assert sha(res["rowptr"]) == str(g[f"rowptr_sha_{s}"])
    assert sha(res["colidx"]) == str(g[f"colidx_sha_{s}"])
    tol = 1e-4 if dt == np.float32 else 1e-10   # f32: sums of up to hundreds of terms, order differs
    assert np.allclose(res["vals"][g[f"sample_idx_{s}"]], g[f"sample_val_{s}"], rtol=tol, atol=0)
    assert np.isclose(res["vals"].astype(np.float64).sum(), float(g[f"val_sum_{s}"]), rtol=tol)


def test_mlp_layer_f32(port, golden_dir):
    """configs[4] shape: act * W^T in f32, within 1e-5 of the reference and of dense f64."""
    g = load(golden_dir, "mlp_expected.npz")
    _, _, ar, ac, av = port.readcoo(os.path.join(golden_dir, "mlp_act.mtx"))
    nr, ncol, br, bc, bv = port.readcoo(os.path.join(golden_dir, "mlp_fc1_weight.mtx"))
    res, _ = spgemm_from_coo(port, 64, 784, 100, (ar, ac, av), (bc, br, bv), np.float32)
    assert res["partials"] == int(g["P"])
    assert np.array_equal(csr_rows(res["rowptr"]), g["rows"])
    assert np.array_equal(res["colidx"], g["cols"])
    assert np.allclose(res["vals"], g["vals"], rtol=1e-5, atol=1e-7)
    dense = g["dense_f64"][g["rows"], g["cols"]]
    assert np.allclose(res["vals"], dense, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("H", [100, 1000])
def test_mlp_layer_full_shape_f32(port, golden_dir, H):
    """configs[4] at its stated shape: act_0 (1024 x 784) * fc1_weight (H x 784)^T, H = 100 (models.py:10-14) and 1000
    (the saved logs), f32: the oracle against the compiled reference's digests and against dense f64."""
    g = load(golden_dir, "mlp_full_expected.npz")
    act, W, Wp, a, b = gen.mlp_layer_operands(H, g[f"thr_{H}"])
    assert sha(act) + sha(W) == str(g[f"in_sha_{H}"]), "generator drifted"
    assert len(a[0]) == int(g[f"act_nnz_{H}"]) and len(b[0]) == int(g[f"w_nnz_{H}"])
    res, _ = spgemm_from_coo(port, 1024, 784, H, a, b, np.float32)
    assert res["partials"] == int(g[f"P_{H}"]) and len(res["colidx"]) == int(g[f"nnzC_{H}"])
    assert sha(res["rowptr"]) == str(g[f"rowptr_sha_{H}"]) and sha(res["colidx"]) == str(g[f"colidx_sha_{H}"])
    scale = float(g[f"val_abs_max_{H}"])   # terms of both signs: 1e-5 of the largest entry
    idx = g[f"sample_idx_{H}"]
    assert np.abs(res["vals"][idx] - g[f"sample_val_{H}"]).max() <= 1e-5 * scale
    assert np.abs(res["vals"][idx] - g[f"sample_dense_f64_{H}"]).max() <= 1e-5 * scale
    assert np.isclose(res["vals"].astype(np.float64).sum(), float(g[f"val_sum_{H}"]), rtol=1e-5, atol=1e-5 * scale)


def test_kslab_decomposition(port):
    """The outer product is k-separable: slab results sum to the full product."""
    n, rows, cols, vals = gen.rmat_coo(8, 8, "mild", seed=3)
    rc, apos, aidx, aval = port.coo2csr(True, n, rows, cols, vals)
    rc, bpos, bidx, bval = port.coo2csr(False, n, rows, cols, vals)
    full = port.spgemm(n, n, n, apos, aidx, aval, bpos, bidx, bval)
    import scipy.sparse as sp
    acc = sp.csr_matrix((n, n))
    P = 0
    for k0, k1 in ((0, 100), (100, 101), (101, n)):
        r = port.spgemm(n, n, n, apos, aidx, aval, bpos, bidx, bval, k0, k1)
        acc = acc + sp.csr_matrix((r["vals"], r["colidx"], r["rowptr"]), shape=(n, n))
        P += r["partials"]
    assert P == full["partials"] == port.mulflops(n, apos, bpos)
    ref = sp.csr_matrix((full["vals"], full["colidx"], full["rowptr"]), shape=(n, n))
    assert abs(acc - ref).max() < 1e-9


# ---- the reference's alternative producers and mergers (merge2way / mergeHardware / merge / multHardware,
# ---- csr2compact / compactMulcsr / csc2rawcompact), compiled from the reference's own text (oracle/Makefile) ----------
def _variant_inputs(golden_dir, port, case, dt):
    if case == "c1":
        _, _, ar, ac, av = port.readcoo(os.path.join(golden_dir, "c1_A.mtx"))
        _, _, br, bc, bv = port.readcoo(os.path.join(golden_dir, "c1_B.mtx"))
        return 64, 64, 64, (ar, ac, av), (bc, br, bv)   # A * B^T, the CLI's product
    if case == "rect":
        g = load(golden_dir, "edges_expected.npz")
        return 5, 7, 3, (g["rect_a_rows"], g["rect_a_cols"], g["rect_a_vals"]), (g["rect_b_rows"], g["rect_b_cols"], g["rect_b_vals"])
    if case == "cancel":
        g = load(golden_dir, "edges_expected.npz")
        return 2, 2, 2, (g["cancel_a_rows"], g["cancel_a_cols"], g["cancel_a_vals"]), (g["cancel_b_rows"], g["cancel_b_cols"], g["cancel_b_vals"])
    if case == "mlp":
        _, _, ar, ac, av = port.readcoo(os.path.join(golden_dir, "mlp_act.mtx"))
        _, _, br, bc, bv = port.readcoo(os.path.join(golden_dir, "mlp_fc1_weight.mtx"))
        return 64, 784, 100, (ar, ac, av), (bc, br, bv)
    n, rows, cols, vals = gen.rmat_coo(10, 16, "g500", seed=1, dtype=dt)
    return n, n, n, (rows, cols, vals), (rows, cols, vals)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("case", ["c1", "rect", "cancel", "mlp", "rmat10"])
def test_reference_variants_agree(port, golden_dir, case, dt):
    """Four formulations inside the reference -- the outer product (cscMulcsr), the outer product through the raw
    compact form, the row-wise product through csr2compact, and the merge tree (multHardware + six layers of merge2way,
    scheduled by merge()) -- and the plain-C restatement all give the same matrix: indices identical, values within
    rounding (the merge tree adds in tree order, deduplicateCOO in std::sort's order, the restatement in ascending k)."""
    M, K, N, a, b = _variant_inputs(golden_dir, port, case, dt)
    want, csx = spgemm_from_coo(port, M, K, N, a, b, dt)
    rows = csr_rows(want["rowptr"])
    ref = orc.ref(dt)
    tol = 2e-5 if dt == np.float32 else 1e-12
    seen = 0
    for name in ref.VARIANTS:
        rc, got = ref.spgemm_variant(name, M, K, *csx)
        if rc == 6:   # the reference's merge() asserts on this input (longest row of A a multiple of 63 beyond 64)
            assert name == "csr2compact+merge"
            continue
        assert rc == 0, (name, rc)
        assert got["partials"] == want["partials"], name
        assert np.array_equal(got["rows"], rows) and np.array_equal(got["cols"], want["colidx"]), name
        scale = np.maximum(np.abs(want["vals"]), np.finfo(dt).tiny)
        if case == "mlp":      # terms of both signs: the error scales with the terms, not with their sum
            scale = np.maximum(scale, np.abs(want["vals"]).max())
        if case == "cancel":   # an exact zero that must stay: 1*2 + 2*(-1)
            assert np.array_equal(got["vals"], want["vals"]), name
        else:
            assert np.all(np.abs(got["vals"] - want["vals"]) <= tol * scale), name
        seen += 1
    assert seen >= 3


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_merge_tree_refuses_what_the_reference_asserts_on(port):
    """merge() starts with mergeK = ways % 63 (SimSpGEMM.cpp:457): 126 streams -> 0 -> its own assert (:483) fails.
    The driver reports that input instead of aborting the test process."""
    K = 126
    a = (np.zeros(K, np.uint32), np.arange(K, dtype=np.uint32), np.ones(K))       # one row of A with 126 non-zeros
    a = (np.concatenate([a[0], [1]]).astype(np.uint32), np.concatenate([a[1], [0]]).astype(np.uint32), np.ones(K + 1))
    b = (np.arange(K, dtype=np.uint32), np.arange(K, dtype=np.uint32) % 5, np.ones(K))
    _, csx = spgemm_from_coo(port, 2, K, 5, a, b, np.float64)
    rc, _ = orc.ref(np.float64).spgemm_variant("csr2compact+merge", 2, K, *csx)
    assert rc == 6
    rc, got = orc.ref(np.float64).spgemm_variant("csr2compact+compactMulcsr", 2, K, *csx)
    assert rc == 0 and got["partials"] == K + 1
