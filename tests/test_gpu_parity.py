"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bar: rowptr / colidx bit-exact.  Values: the merge sums equal keys in staging order (ascending k),
the same order the oracle's stable sort yields, so values are compared BIT-EXACT against the oracle,
and within 1e-6 (f64) / 1e-5 (f32) relative against the reference's goldens (whose std::sort is
unstable, so its summation order is unspecified).
"""
import os

import numpy as np
import pytest

from outerspace_amd import generators as gen

pytestmark = pytest.mark.gpu

RTOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-6}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def csr_rows(rowptr):
    return np.repeat(np.arange(len(rowptr) - 1, dtype=np.uint32), np.diff(rowptr))


def run_both(ctx, port, M, K, N, a, b, dt, **kw):
    """a, b = COO triples.  Returns (gpu CsrResult, oracle dict)."""
    from outerspace_amd import spgemm as S
    acsc = S.coo_to_csc(K, a[0], a[1], a[2].astype(dt))
    bcsr = S.coo_to_csr(K, b[0], b[1], b[2].astype(dt))
    want = port.spgemm(M, K, N, *acsc, *bcsr, *( (kw["k_range"][0], kw["k_range"][1]) if kw.get("k_range") else ()))
    got = ctx.spgemm_csc_csr(M, K, N, *acsc, *bcsr, **kw)
    return got, want


def assert_same(got, want, exact_vals=True):
    assert got.info["partials"] == want["partials"]
    assert np.array_equal(got.rowptr, want["rowptr"])
    assert np.array_equal(got.colidx, want["colidx"])
    if exact_vals:
        assert np.array_equal(got.vals, want["vals"])
    else:
        assert np.allclose(got.vals, want["vals"], rtol=RTOL[np.dtype(got.dtype)], atol=0)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_c1_mtx_cli_flow(ctx, golden_dir, dt):
    """BASELINE configs[0] through the reference CLI's data flow (two .mtx, A * B^T)."""
    g = load(golden_dir, "c1_expected.npz")
    res = ctx.spgemm_mtx(os.path.join(golden_dir, "c1_A.mtx"), os.path.join(golden_dir, "c1_B.mtx"), True, dt)
    s = np.dtype(dt).name
    assert res.info["partials"] == int(g["P"]) and res.nnz == len(g[f"rows_{s}"])
    assert np.array_equal(csr_rows(res.rowptr), g[f"rows_{s}"])
    assert np.array_equal(res.colidx, g[f"cols_{s}"])
    assert np.allclose(res.vals, g[f"vals_{s}"], rtol=RTOL[np.dtype(dt)], atol=0)
    # and A * B (no transpose workaround)
    if dt == np.float64:
        res = ctx.spgemm_mtx(os.path.join(golden_dir, "c1_A.mtx"), os.path.join(golden_dir, "c1_B.mtx"), False, dt)
        assert np.array_equal(csr_rows(res.rowptr), g["nt_rows"]) and np.array_equal(res.colidx, g["nt_cols"])
        assert np.allclose(res.vals, g["nt_vals"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_golden_edges(ctx, port, golden_dir, dt):
    g = load(golden_dir, "edges_expected.npz")
    s = np.dtype(dt).name
    a = (g["rect_a_rows"], g["rect_a_cols"], g["rect_a_vals"])
    b = (g["rect_b_rows"], g["rect_b_cols"], g["rect_b_vals"])
    got, want = run_both(ctx, port, 5, 7, 3, a, b, dt)
    assert_same(got, want)
    assert np.array_equal(csr_rows(got.rowptr), g[f"rect_rows_{s}"])
    assert np.array_equal(got.colidx, g[f"rect_cols_{s}"])
    assert np.allclose(got.vals, g[f"rect_vals_{s}"], rtol=RTOL[np.dtype(dt)], atol=0)


def test_cancellation_keeps_zero(ctx, port, golden_dir):
    g = load(golden_dir, "edges_expected.npz")
    a = (g["cancel_a_rows"], g["cancel_a_cols"], g["cancel_a_vals"])
    b = (g["cancel_b_rows"], g["cancel_b_cols"], g["cancel_b_vals"])
    got, want = run_both(ctx, port, 2, 2, 2, a, b, np.float64)
    assert_same(got, want)
    assert np.array_equal(got.vals, g["cancel_vals"]) and 0.0 in got.vals


def test_empty_operands(ctx, port):
    e = (np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0))
    one = (np.array([1], np.uint32), np.array([2], np.uint32), np.array([3.0]))
    for a, b in ((e, e), (one, e), (e, one)):
        got, want = run_both(ctx, port, 4, 5, 6, a, (b[0], b[1], b[2]), np.float64)
        assert got.nnz == 0 and np.array_equal(got.rowptr, np.zeros(5, np.int64))
        assert_same(got, want)
    # k with a non-empty column of A but an empty row of B
    a = (np.array([0, 3], np.uint32), np.array([1, 1], np.uint32), np.array([1.0, 2.0]))
    b = (np.array([0], np.uint32), np.array([0], np.uint32), np.array([5.0]))
    got, want = run_both(ctx, port, 4, 5, 6, a, b, np.float64)
    assert got.nnz == 0
    assert_same(got, want)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(37, 53, 29, 0.2), (300, 200, 500, 0.05), (1, 64, 1, 0.5), (64, 1, 64, 0.7)])
def test_random_rectangular(ctx, port, dt, shape):
    M, K, N, dens = shape
    a = gen.random_coo(M, K, dens, seed=1, dtype=dt)
    b = gen.random_coo(K, N, dens, seed=2, dtype=dt)
    got, want = run_both(ctx, port, M, K, N, a, b, dt)
    assert_same(got, want)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_rmat10_golden_digest(ctx, golden_dir, dt):
    """Skewed self-product against the reference's digests (heavy rows -> global-sort path)."""
    import hashlib
    from outerspace_amd import spgemm as S
    g = load(golden_dir, "rmat10_expected.npz")
    s = np.dtype(dt).name
    n, rows, cols, vals = gen.rmat_coo(10, 16, "g500", seed=1, dtype=dt)
    res = ctx.spgemm_csc_csr(n, n, n, *S.coo_to_csc(n, rows, cols, vals), *S.coo_to_csr(n, rows, cols, vals))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert res.info["partials"] == int(g[f"P_{s}"]) and res.nnz == int(g[f"nnzC_{s}"])
    assert sha(res.rowptr) == str(g[f"rowptr_sha_{s}"]) and sha(res.colidx) == str(g[f"colidx_sha_{s}"])
    tol = 1e-4 if dt == np.float32 else 1e-6  # f32 rows sum hundreds of terms in a different order
    assert np.allclose(res.vals[g[f"sample_idx_{s}"]], g[f"sample_val_{s}"], rtol=tol, atol=0)
    assert res.info["heavy_rows"] > 0


@pytest.mark.parametrize("dense", ["1", "0"])
def test_long_rows_split_and_fallback(ctx, port, monkeypatch, dense):
    """Rows far longer than an LDS tile: split by column range; a segment that is still too long (one column hit by
    thousands of partial products) is reduced by dense accumulation when its column range is narrow (hub rows), else by a
    big in-place tile or the global-sort path (OSP_DENSE_SEG=0 sends this one there).  Bit-exact every way."""
    monkeypatch.setenv("OSP_DENSE_SEG", dense)
    rng = np.random.default_rng(3)
    M, K, N = 8, 6000, 64
    # row 0 of A is dense in k; every B row hits column 5 plus two random columns -> column 5 of C[0,:]
    # collects K partial products (> one tile), the other columns a few hundred each
    a_rows = np.concatenate([np.zeros(K, np.uint32), rng.integers(1, M, 400).astype(np.uint32)])
    a_cols = np.concatenate([np.arange(K, dtype=np.uint32), rng.choice(K, 400, replace=False).astype(np.uint32)])
    key = np.unique(a_rows.astype(np.int64) * K + a_cols)
    a = ((key // K).astype(np.uint32), (key % K).astype(np.uint32), rng.uniform(0.5, 1.5, len(key)))
    b_rows = np.repeat(np.arange(K, dtype=np.uint32), 3)
    b_cols = np.stack([np.full(K, 5), rng.integers(6, 35, K), rng.integers(35, N, K)], 1).reshape(-1).astype(np.uint32)
    b = (b_rows, b_cols, rng.uniform(0.5, 1.5, 3 * K))
    for dt in (np.float64, np.float32):
        got, want = run_both(ctx, port, M, K, N, a, b, dt)
        assert got.info["heavy_rows"] >= 1
        assert (got.info["dense_segments"] >= 1) if dense == "1" else (got.info["sorted_segments"] >= 1 and got.info["dense_segments"] == 0)
        assert_same(got, want)


def test_many_long_rows(ctx, port):
    """More long rows than a 16-bit grid dimension holds, all split and merged bit-exactly."""
    n, rows, cols, vals = gen.rmat_coo(17, 16, "mild", seed=2)
    got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float64)
    assert got.info["heavy_rows"] > 10000
    assert_same(got, want)


@pytest.mark.parametrize("row_max", ["0", "6000", None])
def test_long_row_split_kernels_agree(ctx, port, monkeypatch, row_max):
    """Long rows are split either by one workgroup per row or by one workgroup per 4096-entry stretch
    (rows above 2^18 partial products); OSP_SPLIT_ROW_MAX moves the boundary so both run here: all rows
    on the per-stretch path, a mix, all on the per-row path.  Same bits every time."""
    if row_max is not None:
        monkeypatch.setenv("OSP_SPLIT_ROW_MAX", row_max)
    for dt in (np.float64, np.float32):
        n, rows, cols, vals = gen.rmat_coo(13, 16, "g500", seed=5, dtype=dt)
        got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt)
        assert got.info["heavy_rows"] > 100
        assert np.array_equal(got.rowptr, want["rowptr"]) and np.array_equal(got.colidx, want["colidx"])
        assert np.array_equal(got.vals, want["vals"])


@pytest.mark.parametrize("preset,scale", [("uniform", 12), ("mild", 12), ("g500", 12)])
def test_rmat_vs_oracle(ctx, port, preset, scale):
    n, rows, cols, vals = gen.rmat_coo(scale, 16, preset, seed=1)
    got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float64)
    assert_same(got, want)


def test_panels_and_kshards(ctx, port):
    """Row-panel tiling (small staging capacity) and k-sharding give the same CSR."""
    n, rows, cols, vals = gen.rmat_coo(11, 8, "mild", seed=5)
    a = b = (rows, cols, vals)
    got1, want = run_both(ctx, port, n, n, n, a, b, np.float64)
    assert got1.info["panels"] == 1
    got, _ = run_both(ctx, port, n, n, n, a, b, np.float64, partial_capacity=want["partials"] // 7)
    assert got.info["panels"] >= 7
    assert_same(got, want)
    # k shards: each equals the oracle's slab; their sum (merge_csr_parts) equals the full product
    parts = []
    for k0, k1 in ((0, 700), (700, 701), (701, n)):
        g, w = run_both(ctx, port, n, n, n, a, b, np.float64, k_range=(k0, k1))
        assert_same(g, w)
        parts.append((g.rowptr, g.colidx, g.vals))
    merged = ctx.merge_csr_parts(n, n, parts)
    assert np.array_equal(merged.rowptr, want["rowptr"]) and np.array_equal(merged.colidx, want["colidx"])
    assert np.allclose(merged.vals, want["vals"], rtol=1e-12, atol=0)  # slab sums re-associate


def test_capacity_error(ctx, port):
    from outerspace_amd import spgemm as S
    n, rows, cols, vals = gen.rmat_coo(10, 16, "g500", seed=1)
    with pytest.raises(S.OspError) as ei:
        run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float64, partial_capacity=1 << 10)
    # capacity is clamped to >= 2^20 only when auto; an explicit tiny value must fail loudly
    assert ei.value.status == 7


def test_validation_errors(ctx):
    from outerspace_amd import spgemm as S
    colptr = np.array([0, 2, 3], np.int64)
    good = (colptr, np.array([0, 1, 1], np.uint32), np.array([1.0, 2.0, 3.0]))
    dup = (colptr, np.array([1, 1, 0], np.uint32), np.array([1.0, 2.0, 3.0]))
    uns = (colptr, np.array([1, 0, 0], np.uint32), np.array([1.0, 2.0, 3.0]))
    rng = (colptr, np.array([0, 9, 0], np.uint32), np.array([1.0, 2.0, 3.0]))
    for bad, status in ((dup, 233), (uns, 8), (rng, 6)):
        with pytest.raises(S.OspError) as ei:
            ctx.spgemm_csc_csr(2, 2, 2, *bad, *good)
        assert ei.value.status == status
        with pytest.raises(S.OspError) as ei:
            ctx.spgemm_csc_csr(2, 2, 2, *good, *bad)
        assert ei.value.status == status
    with pytest.raises(S.OspError) as ei:  # reference: assert(csc.pos.size()==csr.pos.size())
        ctx.spgemm_csc_csr(2, 2, 2, *good, np.array([0, 1], np.int64), good[1][:1], good[2][:1])
    assert ei.value.status == 1
    # host-side conversion: duplicate coordinate is the reference's 233
    with pytest.raises(S.OspError) as ei:
        S.coo_to_csr(3, np.array([0, 1, 1], np.uint32), np.array([0, 2, 2], np.uint32), np.array([1.0, 2.0, 3.0]))
    assert ei.value.status == 233


def test_mlp_layer_f32(ctx, golden_dir):
    """BASELINE configs[4] shape: act * W^T in f32 within 1e-5 of the reference."""
    from outerspace_amd import spgemm as S
    g = load(golden_dir, "mlp_expected.npz")
    res = ctx.spgemm_mtx(os.path.join(golden_dir, "mlp_act.mtx"), os.path.join(golden_dir, "mlp_fc1_weight.mtx"),
                         True, np.float32)
    assert res.info["partials"] == int(g["P"])
    assert np.array_equal(csr_rows(res.rowptr), g["rows"]) and np.array_equal(res.colidx, g["cols"])
    assert np.allclose(res.vals, g["vals"], rtol=1e-5, atol=1e-7)
    # dense/scipy entry point beside sparse_util
    W = np.load(os.path.join(golden_dir, "mlp_weight_dense.npz"))["Wp"]
    _, _, ar, ac, av = S.read_mtx(os.path.join(golden_dir, "mlp_act.mtx"))
    import scipy.sparse as sp
    act = sp.csr_matrix((av.astype(np.float32), (ar, ac)), shape=(64, 784))
    out = S.spgemm(act, W, transpose_b=True, ctx=ctx)
    assert abs(out - sp.csr_matrix((g["vals"], (g["rows"], g["cols"])), shape=(64, 100))).max() < 1e-5


@pytest.mark.parametrize("H", [100, 1000])
def test_mlp_layer_full_shape_f32(ctx, port, golden_dir, H):
    """BASELINE configs[4] at its stated shape (batch 1024, H = 100 and 1000), f32: the GPU bit for bit against the oracle,
    within 1e-5 of the compiled reference's sample and of dense f64; then the same layer through the NN glue
    (sparse_linear with bias and ReLU) against dense torch."""
    import hashlib
    import torch
    from outerspace_amd import sparse_util as su
    g = load(golden_dir, "mlp_full_expected.npz")
    act, W, Wp, a, b = gen.mlp_layer_operands(H, g[f"thr_{H}"])
    got, want = run_both(ctx, port, 1024, 784, H, a, b, np.float32)
    assert_same(got, want)
    sha = lambda x: hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()
    assert got.info["partials"] == int(g[f"P_{H}"]) and got.nnz == int(g[f"nnzC_{H}"])
    assert sha(got.rowptr) == str(g[f"rowptr_sha_{H}"]) and sha(got.colidx) == str(g[f"colidx_sha_{H}"])
    scale = float(g[f"val_abs_max_{H}"])
    idx = g[f"sample_idx_{H}"]
    assert np.abs(got.vals[idx] - g[f"sample_val_{H}"]).max() <= 1e-5 * scale
    assert np.abs(got.vals[idx] - g[f"sample_dense_f64_{H}"]).max() <= 1e-5 * scale
    # the layer as the model runs it: relu(act @ W^T + b)
    bias = torch.linspace(-0.05, 0.05, H)
    out = su.sparse_linear(torch.from_numpy(act), torch.from_numpy(Wp), bias, relu=True, ctx=ctx)
    ref = torch.relu(torch.from_numpy(act).double() @ torch.from_numpy(Wp).double().T + bias.double()).numpy()
    assert out.shape == (1024, H) and np.abs(out.toarray() - ref).max() <= 1e-5 * max(scale, 1.0)
    assert out.nnz == int((out.toarray() != 0).sum())   # ReLU re-sparsified: no explicit zeros kept


def test_properties_at_scale(ctx, port):
    """Size-independent checks (the ones the full-size tests rely on), here at a size where the oracle can confirm them:
    sorted unique columns, row sums (C*1 == A*(B*1)), linearity in A's values -- and the oracle's bits."""
    import scipy.sparse as sp
    from outerspace_amd import spgemm as S
    n, rows, cols, vals = gen.rmat_coo(16, 16, "mild", seed=7)
    acsc = S.coo_to_csc(n, rows, cols, vals)
    bcsr = S.coo_to_csr(n, rows, cols, vals)
    res = ctx.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
    rp, ci, cv = res.rowptr, res.colidx, res.vals
    assert rp[0] == 0 and rp[-1] == res.nnz and np.all(np.diff(rp) >= 0)
    d = np.diff(ci.astype(np.int64))
    starts = rp[1:-1][np.diff(rp)[:-1] >= 0]
    inner = np.ones(len(d), bool)
    inner[starts[(starts > 0) & (starts < len(ci))] - 1] = False
    assert np.all(d[inner] > 0), "columns must be strictly ascending inside every row"
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    want_rowsum = A @ (A @ np.ones(n))
    got_rowsum = np.add.reduceat(np.append(cv, 0.0), np.minimum(rp[:-1], len(cv)))
    got_rowsum[np.diff(rp) == 0] = 0.0
    assert np.allclose(got_rowsum, want_rowsum, rtol=1e-9)
    assert res.nnz == (A @ A).nnz  # values are positive: no cancellation, scipy structure agrees
    res2 = ctx.spgemm_csc_csr(n, n, n, acsc[0], acsc[1], 2.0 * acsc[2], *bcsr)
    assert np.array_equal(res2.colidx, ci) and np.array_equal(res2.vals, 2.0 * cv)
    assert_same(res, port.spgemm(n, n, n, *acsc, *bcsr))


def test_cli_reference_call_shape(golden_dir, tmp_path):
    """`osp_spgemm A.mtx B.mtx` mirrors `./simulator A.mtx B.mtx` (SimSpGEMM.cpp:819-894)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "outerspace_amd", "osp_spgemm")
    out = tmp_path / "c.mtx"
    r = subprocess.run([exe, os.path.join(golden_dir, "c1_A.mtx"), os.path.join(golden_dir, "c1_B.mtx"), "--out", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "NCol = 64, NRow = 64, NNZ = 410" in r.stdout          # the reference's (swapped) labels, :866
    assert "mul flops ref = 2692" in r.stdout                      # :891
    from outerspace_amd import spgemm as S
    g = load(golden_dir, "c1_expected.npz")
    nrow, ncol, rr, cc, vv = S.read_mtx(str(out))
    assert (nrow, ncol) == (64, 64) and np.array_equal(rr, g["rows_float32"]) and np.array_equal(cc, g["cols_float32"])
    assert np.allclose(vv, g["vals_float32"], rtol=1e-5, atol=0)
    # duplicate coordinate: the reference dies with an uncaught throw(233)
    dup = tmp_path / "dup.mtx"
    dup.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n2 2 2.0\n2 2 3.0\n")
    r = subprocess.run([exe, str(dup), str(dup)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 233 and "233" in r.stderr


def test_device_pointer_api_with_torch(ctx, port):
    """Operands resident in HBM (torch tensors), result borrowed in place -- the bench / multi-GPU path."""
    import torch
    from outerspace_amd import spgemm as S
    from outerspace_amd.distributed import _as_tensor
    n, rows, cols, vals = gen.rmat_coo(12, 8, "mild", seed=9)
    acsc = S.coo_to_csc(n, rows, cols, vals)
    bcsr = S.coo_to_csr(n, rows, cols, vals)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev) for a in (*acsc, *bcsr)]
    res = ctx.spgemm_csc_csr_device(np.float64, n, n, n, [x.data_ptr() for x in t], validate=True)
    want = port.spgemm(n, n, n, *acsc, *bcsr)
    rp, ci, va = res.device_ptrs()
    got_rp = _as_tensor(rp, n + 1, "<i8", dev, torch.int64).cpu().numpy()
    got_ci = _as_tensor(ci, res.nnz, "<i4", dev, torch.int32).cpu().numpy().view(np.uint32)
    got_va = _as_tensor(va, res.nnz, "<f8", dev, torch.float64).cpu().numpy()
    assert np.array_equal(got_rp, want["rowptr"]) and np.array_equal(got_ci, want["colidx"]) and np.array_equal(got_va, want["vals"])


def test_power_law_web_graph_shape(ctx, port):
    """BASELINE configs[1] shape (web-Google-like: power-law degrees, pattern values = 1.0), reduced size,
    against the oracle (bit-exact) and scipy: structure exact, values exact (small integers in f64)."""
    import scipy.sparse as sp
    from outerspace_amd import spgemm as S
    rng = np.random.default_rng(5)
    n = 60000
    deg = np.minimum((rng.pareto(1.2, n) + 1).astype(np.int64) * 2, 3000)
    rows = np.repeat(np.arange(n), deg)
    cols = (rng.pareto(0.9, len(rows)) * 50).astype(np.int64) % n      # a few very popular targets
    key = np.unique(rows * n + cols)
    rows, cols = (key // n).astype(np.uint32), (key % n).astype(np.uint32)
    vals = np.ones(len(key))                                             # pattern file -> 1.0 (SimSpGEMM.cpp:92-93)
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    res = ctx.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
    assert_same(res, port.spgemm(n, n, n, *acsc, *bcsr))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    want = (A @ A).tocsr()
    want.sort_indices()
    assert np.array_equal(res.rowptr, want.indptr) and np.array_equal(res.colidx, want.indices.astype(np.uint32))
    assert np.array_equal(res.vals, want.data)
    assert res.info["heavy_rows"] > 0


def test_merge_parts_with_empty_and_single(ctx):
    from outerspace_amd import spgemm as S
    import scipy.sparse as sp
    rng = np.random.default_rng(1)
    mats = [sp.random(50, 40, 0.1, random_state=rng, format="csr"), sp.csr_matrix((50, 40)),
            sp.random(50, 40, 0.3, random_state=rng, format="csr")]
    for m in mats:
        m.sort_indices()
    parts = [(m.indptr.astype(np.int64), m.indices.astype(np.uint32), m.data) for m in mats]
    got = ctx.merge_csr_parts(50, 40, parts).to_scipy()
    want = mats[0] + mats[1] + mats[2]
    assert abs(got - want).max() < 1e-12 and got.nnz == want.nnz
    one = ctx.merge_csr_parts(50, 40, parts[2:])
    assert np.array_equal(one.rowptr, parts[2][0]) and np.array_equal(one.colidx, parts[2][1]) and np.array_equal(one.vals, parts[2][2])


def test_f32_skewed_bit_exact(ctx, port):
    n, rows, cols, vals = gen.rmat_coo(13, 16, "g500", seed=4, dtype=np.float32)
    got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float32)
    assert got.info["heavy_rows"] > 0
    assert_same(got, want)


def test_huge_column_space_and_many_empty_rows(ctx, port):
    """N close to 2^32 (32 key bits for the column alone: one row per tile, no row bits) and an output with
    almost only empty rows (tiles of 256 empty rows)."""
    rng = np.random.default_rng(11)
    M, K, N = 300, 64, 4_000_000_000
    a = gen.random_coo(M, K, 0.05, seed=3)
    b_rows = np.repeat(np.arange(K, dtype=np.uint32), 20)
    b_cols = np.sort(rng.integers(0, N, (K, 20), dtype=np.int64), axis=1)
    b_cols[:, 1] = b_cols[:, 0] + 1           # adjacent columns
    b_cols[:, 19] = N - 1                      # every B row hits the last column -> duplicates to sum
    b_cols = np.sort(b_cols, axis=1)
    keep = np.ones(b_cols.shape, bool)
    keep[:, 1:] = b_cols[:, 1:] != b_cols[:, :-1]
    b = (b_rows[keep.reshape(-1)], b_cols.reshape(-1)[keep.reshape(-1)].astype(np.uint32), rng.uniform(0.5, 1.5, int(keep.sum())))
    got, want = run_both(ctx, port, M, K, N, a, b, np.float64)
    assert_same(got, want)
    assert got.colidx.max() == N - 1
    # 2 million output rows, 500 non-zeros of A
    M2 = 2_000_000
    a2 = (np.sort(rng.choice(M2, 500, replace=False)).astype(np.uint32), rng.integers(0, K, 500).astype(np.uint32),
          rng.uniform(0.5, 1.5, 500))
    b2 = gen.random_coo(K, 1000, 0.02, seed=5)
    got, want = run_both(ctx, port, M2, K, 1000, a2, b2, np.float64)
    assert_same(got, want)


def test_device_ingest_coo(ctx, port):
    """osp_spgemm_coo: COO in any order -> CSC/CSR on the device (coo2csr, SimSpGEMM.cpp:102-152) -> product."""
    from outerspace_amd import spgemm as S
    rng = np.random.default_rng(2)
    n, rows, cols, vals = gen.rmat_coo(12, 12, "g500", seed=6)
    pa, pb = rng.permutation(len(rows)), rng.permutation(len(rows))           # shuffled input order
    a = (rows[pa], cols[pa], vals[pa])
    b = (rows[pb], cols[pb], vals[pb])
    got = ctx.spgemm_coo(n, n, n, a, b)
    want = port.spgemm(n, n, n, *S.coo_to_csc(n, rows, cols, vals), *S.coo_to_csr(n, rows, cols, vals))
    assert_same(got, want)
    assert got.info["ms_ingest"] > 0
    # rectangular, f32, with the golden edge case
    e = load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"), "edges_expected.npz")
    a = (e["rect_a_rows"], e["rect_a_cols"], e["rect_a_vals"].astype(np.float32))
    b = (e["rect_b_rows"], e["rect_b_cols"], e["rect_b_vals"].astype(np.float32))
    got = ctx.spgemm_coo(5, 7, 3, a, b)
    assert np.array_equal(csr_rows(got.rowptr), e["rect_rows_float32"]) and np.array_equal(got.colidx, e["rect_cols_float32"])
    assert np.allclose(got.vals, e["rect_vals_float32"], rtol=1e-5, atol=0)
    # duplicate coordinate -> 233 (reference: throw(233)); index out of range -> 6
    with pytest.raises(S.OspError) as ei:
        ctx.spgemm_coo(3, 3, 3, (e["dup_rows"], e["dup_cols"], e["dup_vals"]), (e["dup_rows"][:1], e["dup_cols"][:1], e["dup_vals"][:1]))
    assert ei.value.status == 233
    with pytest.raises(S.OspError) as ei:
        ctx.spgemm_coo(3, 3, 3, (np.array([0, 7], np.uint32), np.array([0, 1], np.uint32), np.array([1.0, 2.0])),
                       (np.array([0], np.uint32), np.array([0], np.uint32), np.array([1.0])))
    assert ei.value.status == 6
    # empty operands
    z = (np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0))
    assert ctx.spgemm_coo(4, 5, 6, z, z).nnz == 0


def test_row_shards_tile_the_product(ctx, port):
    """Row-sharded multi-GPU mode: G results with disjoint, contiguous row ranges that concatenate to the full CSR
    (bit-exact), ranges balanced by estimated work and derived identically by every rank."""
    n, rows, cols, vals = gen.rmat_coo(13, 12, "mild", seed=8)
    got_full, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float64)
    from outerspace_amd import spgemm as S
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    for G in (2, 3, 8):
        end, P = 0, 0
        for i in range(G):
            r = ctx.spgemm_csc_csr(n, n, n, *acsc, *bcsr, row_shard=(i, G), partial_capacity=want["partials"] // 5)
            r0, r1 = r.info["row_begin"], r.info["row_end"]
            assert r0 == end and r.shape[0] == r1 - r0
            lo, hi = want["rowptr"][r0], want["rowptr"][r1]
            assert np.array_equal(r.rowptr, want["rowptr"][r0:r1 + 1] - lo)
            assert np.array_equal(r.colidx, want["colidx"][lo:hi]) and np.array_equal(r.vals, want["vals"][lo:hi])
            assert r.info["partials"] <= want["partials"] / G * 2.0 + 5000  # balanced by estimated work (long rows weigh more)
            end, P = r1, P + r.info["partials"]
        assert end == n and P == want["partials"]


def test_streamed_panels_equal_the_resident_product(ctx, port):
    """osp_spgemm_csc_csr_panels: panels arrive in row order, tile the rows exactly, and put together are the CSR the
    resident call returns (which is the oracle's); also as one row shard of three, and with an empty product."""
    import torch
    from outerspace_amd import spgemm as S
    from outerspace_amd.distributed import _as_tensor
    n, rows, cols, vals = gen.rmat_coo(12, 16, "mild", seed=4)
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    want = port.spgemm(n, n, n, *acsc, *bcsr)
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(a.astype(np.int32) if a.dtype == np.uint32 else a).to(dev) for a in (*acsc, *bcsr)]
    ptrs = [x.data_ptr() for x in t]

    def collect(**kw):
        got = []

        def on_panel(p):
            nr = p["row_end"] - p["row_begin"]
            rp = _as_tensor(p["rowptr"], nr + 1, "<i8", dev, torch.int64).cpu().numpy()
            ci = _as_tensor(p["colidx"], p["nnz"], "<i4", dev, torch.int32).cpu().numpy().view(np.uint32)
            cv = _as_tensor(p["vals"], p["nnz"], "<f8", dev, torch.float64).cpu().numpy()
            got.append((dict(p), rp, ci, cv))
        info = ctx.spgemm_csc_csr_panels(np.float64, n, n, n, ptrs, on_panel, **kw)
        return info, got

    P = want["partials"]
    for cap in (0, P // 5 + 1):   # one panel; several panels
        info, got = collect(partial_capacity=cap)
        assert info["nnz_c"] == len(want["colidx"]) and info["partials"] == P and info["panels"] == len(got)
        assert (cap == 0) == (len(got) == 1)
        assert got[0][0]["row_begin"] == 0 and got[-1][0]["row_end"] == n
        assert all(a[0]["row_end"] == b[0]["row_begin"] for a, b in zip(got, got[1:]))
        assert [g[0]["index"] for g in got] == list(range(len(got))) and all(g[0]["count"] == len(got) for g in got)
        for d, rp, ci, cv in got:
            lo, hi = want["rowptr"][d["row_begin"]], want["rowptr"][d["row_end"]]
            assert rp[0] == 0 and rp[-1] == d["nnz"] == hi - lo
            assert np.array_equal(rp, want["rowptr"][d["row_begin"]:d["row_end"] + 1] - lo)
            assert np.array_equal(ci, want["colidx"][lo:hi]) and np.array_equal(cv, want["vals"][lo:hi])
    # a row shard, streamed
    info, got = collect(partial_capacity=P // 7 + 1, row_shard=(1, 3))
    r0, r1 = info["row_begin"], info["row_end"]
    assert 0 < r0 < r1 < n and got[0][0]["row_begin"] == r0 and got[-1][0]["row_end"] == r1
    assert sum(g[0]["nnz"] for g in got) == want["rowptr"][r1] - want["rowptr"][r0] == info["nnz_c"]
    # an exception in the callback aborts the product and comes back out
    def boom(p):
        raise KeyError("stop")
    with pytest.raises(KeyError):
        ctx.spgemm_csc_csr_panels(np.float64, n, n, n, ptrs, boom)
    # empty product: one empty panel covering every row
    z = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    e = torch.zeros(1, dtype=torch.float64, device=dev)
    seen = []
    info = ctx.spgemm_csc_csr_panels(np.float64, n, n, n, [z.data_ptr(), e.data_ptr(), e.data_ptr()] * 2, lambda p: seen.append(dict(p)))
    assert info["nnz_c"] == 0 and len(seen) == 1 and (seen[0]["row_begin"], seen[0]["row_end"], seen[0]["nnz"]) == (0, n, 0)


# ---- the reference's in-memory layout, the binding a maintainer would add, the buffer pool -------------------------------
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_aos_entry_point_takes_the_reference_layout(ctx, port, dt):
    """osp_spgemm_csc_csr_aos: CSRMatrix{vector<size_t> pos; vector<CSRElement{idx,val}> data} (common.h:10-16,39-47) as
    it stands -- packed 8 / 12-byte records, unsigned 64-bit offsets.  Same bits as the SoA call and the oracle."""
    from outerspace_amd import spgemm as S
    n, rows, cols, vals = gen.rmat_coo(11, 12, "g500", seed=12, dtype=dt)
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    rec = S.aos_dtype(dt)
    assert rec.itemsize == 4 + np.dtype(dt).itemsize
    ad, bd = np.empty(len(rows), rec), np.empty(len(rows), rec)
    ad["idx"], ad["val"], bd["idx"], bd["val"] = acsc[1], acsc[2], bcsr[1], bcsr[2]
    got = ctx.spgemm_csc_csr_aos(n, n, n, acsc[0].astype(np.uint64), ad, bcsr[0].astype(np.uint64), bd)
    want = port.spgemm(n, n, n, *acsc, *bcsr)
    assert_same(got, want)
    # a duplicate inside a column is the reference's 233 here too; an empty product works
    bad = ad.copy()
    j = int(np.flatnonzero(np.diff(acsc[0]) >= 2)[0])
    bad["idx"][acsc[0][j] + 1] = bad["idx"][acsc[0][j]]
    with pytest.raises(S.OspError) as ei:
        ctx.spgemm_csc_csr_aos(n, n, n, acsc[0].astype(np.uint64), bad, bcsr[0].astype(np.uint64), bd)
    assert ei.value.status == 233
    z = np.zeros(5, np.uint64)
    assert ctx.spgemm_csc_csr_aos(3, 4, 6, z, np.empty(0, rec), z, np.empty(0, rec)).nnz == 0


@pytest.mark.parametrize("sfx,flag", [("f32", []), ("f64", [])])
def test_reference_code_calls_the_library_through_the_binding(golden_dir, sfx, flag):
    """oracle/_ref/binding_demo_*: the reference's own readcoo / coo2csr / cscMulcsr / deduplicateCOO (compiled from its
    text in the build container) next to integration/simspgemm_gpu_binding.h -- the drop-in of INTEGRATION.md section 2 --
    on the configs[0] files: the GPU result must match the reference's CPU result."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", f"binding_demo_{sfx}")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/binding_demo_* not built (needs /root/reference at build time)")
    for a, b in (("c1_A.mtx", "c1_B.mtx"), ("mlp_act.mtx", "mlp_fc1_weight.mtx")):
        r = subprocess.run([exe, os.path.join(golden_dir, a), os.path.join(golden_dir, b)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "MATCH:" in r.stdout, r.stdout + r.stderr
    assert "mul flops ref = 39574" in r.stdout   # mulflops_ref of the MLP layer (mlp_expected.npz: P)


def test_pool_buffers_for_the_exchange(ctx):
    """osp_context_alloc / osp_context_free: device memory out of the context's pool, usable from torch zero-copy, and
    handed back to the pool (the multi-GPU exchange receives into such buffers)."""
    import torch
    from outerspace_amd.distributed import _as_tensor
    dev = torch.device("cuda", 0)
    p = ctx.alloc(1 << 20)
    assert p and p % 256 == 0
    t = _as_tensor(p, 1 << 18, "<i4", dev, torch.int32)
    t.copy_(torch.arange(1 << 18, dtype=torch.int32, device=dev))
    assert int(t.sum()) == (1 << 18) * ((1 << 18) - 1) // 2
    ctx.free(p)
    q = ctx.alloc(1 << 20)     # comes out of the pool again (best fit: the block just released, or one like it)
    assert q and q % 256 == 0
    ctx.free(q)
    assert ctx.alloc(0)        # a zero-byte request still yields a valid (minimal) block


# ---- BASELINE configs[1] and configs[2] at their full sizes ----------------------------------------------------------------
_FULL = {}


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("osp_bench", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_webgoogle_shape_full_size_vs_oracle(ctx, port):
    """configs[1] at its full size: the 916 428-vertex web-Google-shaped pattern matrix of bench.py (nnz ~5.2 M, P ~54 M,
    nnz(C) ~43 M; the real SuiteSparse file is not available offline), whole self-product against the plain-C oracle:
    rowptr / colidx / values bit for bit."""
    import torch
    from outerspace_amd.distributed import _as_tensor
    dev = torch.device("cuda", 0)
    if "web" not in _FULL:
        n, csr, csc = _bench_module().webgoogle_device(1, dev, torch.float64)
        host = [t.cpu().numpy() for t in (*csc, *csr)]
        host = [a.view(np.uint32) if a.dtype == np.int32 else a for a in host]
        _FULL["web"] = (n, csr, csc, port.spgemm(n, n, n, *host))
    n, csr, csc, want = _FULL["web"]
    assert n == 916428 and want["partials"] > 50_000_000
    res = ctx.spgemm_csc_csr_device(np.float64, n, n, n, [t.data_ptr() for t in (*csc, *csr)], validate=True)
    assert res.info["partials"] == want["partials"] and res.nnz == len(want["colidx"])
    rp, ci, va = res.device_ptrs()
    assert torch.equal(_as_tensor(rp, n + 1, "<i8", dev, torch.int64), torch.from_numpy(want["rowptr"]).to(dev))
    assert torch.equal(_as_tensor(ci, res.nnz, "<i4", dev, torch.int32), torch.from_numpy(want["colidx"].view(np.int32)).to(dev))
    assert torch.equal(_as_tensor(va, res.nnz, "<f8", dev, torch.float64), torch.from_numpy(want["vals"]).to(dev))
    res.close()


@pytest.mark.parametrize("tname", ["f64", "f32"])
def test_lattice_shape_vs_oracle(ctx, port, tname):
    """The cage15-shaped matrix of bench.py at a reduced size (40^3 = 64 000 vertices, ~1.2 M entries, ~24 M partial products,
    about half of them duplicates -- R-MAT products hardly compress): the whole square against the plain-C oracle, bit for bit."""
    import torch
    from outerspace_amd.distributed import _as_tensor
    dev = torch.device("cuda", 0)
    tdt, ndt, vfmt = (torch.float64, np.float64, "<f8") if tname == "f64" else (torch.float32, np.float32, "<f4")
    n, csr, csc = _bench_module().cage15_device(3, dev, tdt, side=40)
    host = [t.cpu().numpy() for t in (*csc, *csr)]
    host = [a.view(np.uint32) if a.dtype == np.int32 else a for a in host]
    want = port.spgemm(n, n, n, *host)
    assert n == 64000 and want["partials"] > 2 * len(want["colidx"]) * 0.8   # it does compress
    res = ctx.spgemm_csc_csr_device(ndt, n, n, n, [t.data_ptr() for t in (*csc, *csr)], validate=True)
    assert res.info["partials"] == want["partials"] and res.nnz == len(want["colidx"])
    rp, ci, va = res.device_ptrs()
    assert torch.equal(_as_tensor(rp, n + 1, "<i8", dev, torch.int64), torch.from_numpy(want["rowptr"]).to(dev))
    assert torch.equal(_as_tensor(ci, res.nnz, "<i4", dev, torch.int32), torch.from_numpy(want["colidx"].view(np.int32)).to(dev))
    assert torch.equal(_as_tensor(va, res.nnz, vfmt, dev, tdt), torch.from_numpy(want["vals"]).to(dev))
    res.close()


def test_stream_copy_probe(_ctx_shared):
    """osp_stream_copy_probe (bench.py's roofline.peak_measured): a plain copy on the context's stream, read + written bytes per
    second, on buffers far beyond the last-level cache (2 GiB each).  The number is a measurement, not a constant of one SKU:
    it must be positive, finite and not absurdly small; bad arguments are refused."""
    import ctypes as C
    import math
    from outerspace_amd import _lib
    g = _ctx_shared.stream_copy_gbps(2 << 30, 3)
    assert math.isfinite(g) and g > 100.0, g
    out = C.c_double()
    assert _lib.lib().osp_stream_copy_probe(_ctx_shared._h, 16, 1, C.byref(out)) == _lib.ERR_ARG
    assert _lib.lib().osp_stream_copy_probe(_ctx_shared._h, 1 << 20, 0, C.byref(out)) == _lib.ERR_ARG


@pytest.mark.parametrize("min_waste", ["0", "default"])
def test_compressing_product_keeps_or_copies_its_bound_sized_arrays(port, monkeypatch, min_waste):
    """A product that compresses leaves the tails of its bound-sized arrays unused.  The copy to exact-size arrays is made only
    where the tails are worth it (more than a tenth of the device's memory; OSP_COMPACT_MIN_WASTE=0: always): same result
    either way, and the info says which happened (output_slack_bytes, ms_compact)."""
    import torch
    from outerspace_amd import spgemm as S
    if min_waste != "default":
        monkeypatch.setenv("OSP_COMPACT_MIN_WASTE", min_waste)
    dev = torch.device("cuda", 0)
    n, csr, csc = _bench_module().cage15_device(5, dev, torch.float64, side=24)
    host = [t.cpu().numpy() for t in (*csc, *csr)]
    host = [a.view(np.uint32) if a.dtype == np.int32 else a for a in host]
    want = port.spgemm(n, n, n, *host)
    with S.Context(0) as c2:
        res = c2.spgemm_csc_csr_device(np.float64, n, n, n, [t.data_ptr() for t in (*csc, *csr)])
        assert np.array_equal(res.rowptr, want["rowptr"]) and np.array_equal(res.colidx, want["colidx"]) and np.array_equal(res.vals, want["vals"])
        if min_waste == "0":
            assert res.info["output_slack_bytes"] == 0 and res.info["ms_compact"] > 0
        else:
            assert res.info["output_slack_bytes"] > 0 and res.info["ms_compact"] == 0
        res.close()


def test_rmat22_full_size_properties_and_slab_parity(ctx, port):
    """configs[2] at its full size (R-MAT scale 22, edge factor 16, (a,b,c,d) = (.45,.22,.22,.11), seed 1: nnz 67 M,
    P = 1.19e10, nnz(C) = 1.15e10 -- 138 GB of CSR, far beyond what the oracle can form):
      * size-independent properties of the WHOLE result, checked on the device: rowptr monotone and ending at nnz,
        columns strictly ascending inside every row, row sums C*1 = A*(B*1), total 1^T C 1 = (1^T A)(B 1);
      * linearity: the product of (2A) and B, streamed panel by panel, is exactly 2C with the same structure;
      * parity proper on two k-slabs (the unit a k-shard computes), bit for bit against the oracle: the hub column k = 0
        alone (9.9e7 partial products, no duplicates) and 77 k columns from the middle of the range (5e7 partial products);
      * parity proper on a ROW slab of the WHOLE product -- rows [0, r1) with 5e7 partial products over all k, the hub row and
        planned long rows among them -- bit for bit against the oracle (entry counts per row, columns, values), and the
        column sums of the whole result against B^T (A^T 1)."""
    import sys
    import time
    import torch
    from outerspace_amd.distributed import _as_tensor
    t_start = time.time()

    def note(what):  # progress on the real stderr: a long test must not look hung
        print(f"[rmat22 {time.time() - t_start:6.1f} s] {what}", file=sys.__stderr__, flush=True)
    dev = torch.device("cuda", 0)
    if "rmat22" not in _FULL:
        _FULL["rmat22"] = _bench_module().rmat_device(22, 16, gen.RMAT_PRESETS["mild"], 1, dev, torch.float64)
    n, csr, csc = _FULL["rmat22"]
    ptrs = [t.data_ptr() for t in (*csc, *csr)]
    # ---- slab parity against the oracle: the hub column alone (k = 0: one outer product, no duplicate coordinates, the
    # longest chunks there are) and a slab from the middle of the k range (tens of thousands of columns, many duplicates) ----
    w = (csc[0][1:] - csc[0][:-1]) * (csr[0][1:] - csr[0][:-1])
    cum = torch.cumsum(w, 0)
    km = n // 3
    k_hi = int(torch.searchsorted(cum, (cum[km - 1] + 50_000_000).reshape(1))[0]) + 1
    for k0, k1 in ((0, 1), (km, k_hi)):
        ea0, ea1, eb0, eb1 = int(csc[0][k0]), int(csc[0][k1]), int(csr[0][k0]), int(csr[0][k1])
        host = [(csc[0][k0:k1 + 1] - ea0).cpu().numpy(), csc[1][ea0:ea1].cpu().numpy().view(np.uint32), csc[2][ea0:ea1].cpu().numpy(),
                (csr[0][k0:k1 + 1] - eb0).cpu().numpy(), csr[1][eb0:eb1].cpu().numpy().view(np.uint32), csr[2][eb0:eb1].cpu().numpy()]
        note(f"k-slab [{k0},{k1}): {int(w[k0:k1].sum())} partial products for the oracle")
        want = port.spgemm(n, k1 - k0, n, *host)
        note("oracle done")
        res = ctx.spgemm_csc_csr_device(np.float64, n, n, n, ptrs, k_range=(k0, k1))
        assert res.info["partials"] == want["partials"] == int(w[k0:k1].sum()) and res.nnz == len(want["colidx"])
        rp, ci, va = res.device_ptrs()
        assert torch.equal(_as_tensor(rp, n + 1, "<i8", dev, torch.int64), torch.from_numpy(want["rowptr"]).to(dev))
        assert torch.equal(_as_tensor(ci, res.nnz, "<i4", dev, torch.int32), torch.from_numpy(want["colidx"].view(np.int32)).to(dev))
        assert torch.equal(_as_tensor(va, res.nnz, "<f8", dev, torch.float64), torch.from_numpy(want["vals"]).to(dev))
        res.close()
        del want, host
    del cum
    # ---- the whole product ----
    note("slab parity ok; whole product")
    res = ctx.spgemm_csc_csr_device(np.float64, n, n, n, ptrs)
    nnz = res.nnz
    note(f"whole product done: nnz {nnz}, {res.info['ms_total']:.0f} ms")
    assert res.info["partials"] == int(w.sum()) and nnz > 11_000_000_000
    rp, ci, va = res.device_ptrs()
    rowptr = _as_tensor(rp, n + 1, "<i8", dev, torch.int64)
    colidx = _as_tensor(ci, nnz, "<i4", dev, torch.int32)
    vals = _as_tensor(va, nnz, "<f8", dev, torch.float64)
    assert int(rowptr[0]) == 0 and int(rowptr[-1]) == nnz and bool((rowptr[1:] >= rowptr[:-1]).all())
    b1 = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(
        0, torch.repeat_interleave(torch.arange(n, device=dev), csr[0][1:] - csr[0][:-1]), csr[2])          # B * 1
    want_rowsum = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, csc[1].long(), csc[2] * torch.repeat_interleave(
        b1, csc[0][1:] - csc[0][:-1]))                                                                       # A * (B * 1)
    got_rowsum = torch.zeros(n, dtype=torch.float64, device=dev)
    ctx.trim()   # the checks below allocate through torch: give it the staging memory the product has released
    # walk the result in blocks of whole rows holding about 2^27 entries: per-row sums by segment reduction (an index_add_
    # of sorted indices would serialise on one address per row), column order by comparing neighbours
    CH = 1 << 27
    cuts = torch.searchsorted(rowptr, torch.arange(0, nnz + CH, CH, device=dev).clamp_(max=nnz)).tolist()
    cuts = sorted(set([0] + [min(int(x), n) for x in cuts] + [n]))
    for bi, (ra, rb) in enumerate(zip(cuts[:-1], cuts[1:])):
        lo, hi = int(rowptr[ra]), int(rowptr[rb])
        if hi == lo:
            continue
        lengths = rowptr[ra + 1:rb + 1] - rowptr[ra:rb]
        got_rowsum[ra:rb] = torch.segment_reduce(vals[lo:hi], "sum", lengths=lengths, unsafe=True)
        rows = torch.repeat_interleave(torch.arange(ra, rb, device=dev, dtype=torch.int32), lengths)
        c = colidx[lo:hi]   # columns < 2^22 here: int32 compares as unsigned
        bad = (c[1:] <= c[:-1]) & (rows[1:] == rows[:-1])
        assert not bool(bad.any()), "columns must ascend strictly inside every row"
        del rows, c, bad, lengths
        if bi % 16 == 0:
            note(f"walked {hi} of {nnz} entries")
    note("structure and row sums walked")
    assert torch.allclose(got_rowsum, want_rowsum, rtol=1e-9, atol=0)
    assert abs(float(got_rowsum.sum()) - float(want_rowsum.sum())) <= 1e-9 * float(want_rowsum.sum())
    del got_rowsum, want_rowsum, b1
    # ---- column sums: C^T 1 = B^T (A^T 1) (a value moved to another column of its row keeps every row sum) ----
    a1 = torch.segment_reduce(csc[2], "sum", lengths=csc[0][1:] - csc[0][:-1], unsafe=True)                  # A^T * 1, per k
    want_colsum = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(
        0, csr[1].long(), csr[2] * torch.repeat_interleave(a1, csr[0][1:] - csr[0][:-1]))
    got_colsum = torch.zeros(n, dtype=torch.float64, device=dev)
    for s0 in range(0, nnz, CH):
        s1 = min(s0 + CH, nnz)
        got_colsum.index_add_(0, colidx[s0:s1].long(), vals[s0:s1])
    assert torch.allclose(got_colsum, want_colsum, rtol=1e-9, atol=0)
    del got_colsum, want_colsum, a1
    note("column sums compared")
    # ---- parity proper on a ROW slab of the whole product (all k, some rows): the long-row merge at full size, bit for bit.
    # Rows [0, r1) holding about 5e7 partial products: in R-MAT the heaviest rows come first -- the hub row (beyond the
    # planner), rows of tens of ranges and rows of two.  The oracle multiplies A restricted to those rows. ----
    kcol = torch.repeat_interleave(torch.arange(n, device=dev), csc[0][1:] - csc[0][:-1])                    # column of every entry of A
    U = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, csc[1].long(), (csr[0][1:] - csr[0][:-1])[kcol])
    r1 = int(torch.searchsorted(torch.cumsum(U, 0), torch.tensor([50_000_000], device=dev))[0]) + 1
    keep = csc[1].long() < r1
    a_colptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    a_colptr[1:] = torch.cumsum(torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, kcol[keep], torch.ones_like(kcol[keep])), 0)
    slab_p = int(U[:r1].sum())
    kinds = (int((U[:r1] > 131072).sum()), int(((U[:r1] > 1536) & (U[:r1] <= 131072)).sum()), int((U[:r1] <= 1536).sum()))
    note(f"row slab [0,{r1}): {slab_p} partial products ({kinds[0]} rows beyond the planner, {kinds[1]} planned long rows, {kinds[2]} short) for the oracle")
    assert kinds[0] >= 1 and kinds[1] >= 1
    want = port.spgemm(n, n, n, a_colptr.cpu().numpy(), csc[1][keep].cpu().numpy().view(np.uint32), csc[2][keep].cpu().numpy(),
                       csr[0].cpu().numpy(), csr[1].cpu().numpy().view(np.uint32), csr[2].cpu().numpy())
    note("oracle done")
    assert want["partials"] == slab_p
    hi = int(rowptr[r1])
    assert torch.equal(rowptr[:r1 + 1], torch.from_numpy(want["rowptr"][:r1 + 1]).to(dev)) and hi == len(want["colidx"])   # per-row entry counts too
    assert torch.equal(colidx[:hi], torch.from_numpy(want["colidx"].view(np.int32)).to(dev))
    assert torch.equal(vals[:hi], torch.from_numpy(want["vals"]).to(dev))
    del want, kcol, U, keep, a_colptr
    note("row slab compared bit for bit")
    # ---- linearity, streamed: (2A) * B == 2 * C exactly, panel by panel against the resident result ----
    a2 = csc[2] * 2.0
    torch.cuda.synchronize()   # the library runs on its own stream
    seen = {"rows": 0, "nnz": 0}

    def on_panel(p):
        lo = int(rowptr[p["row_begin"]])
        nr = p["row_end"] - p["row_begin"]
        prp = _as_tensor(p["rowptr"], nr + 1, "<i8", dev, torch.int64)
        ref = rowptr[p["row_begin"]:p["row_end"] + 1] - lo
        if not torch.equal(prp, ref):   # say where: the first row whose entry count differs
            r = int(torch.nonzero(prp != ref)[0]) - 1
            raise AssertionError(f"panel {p['index']} rows [{p['row_begin']},{p['row_end']}): row {p['row_begin'] + r} holds "
                                 f"{int(prp[r + 1] - prp[r])} entries, resident product {int(ref[r + 1] - ref[r])}")
        assert p["nnz"] == int(rowptr[p["row_end"]]) - lo
        note(f"panel {p['index'] + 1} of {p['count']}")
        for s0 in range(0, p["nnz"], CH):
            s1 = min(s0 + CH, p["nnz"])
            assert torch.equal(_as_tensor(p["colidx"] + 4 * s0, s1 - s0, "<i4", dev, torch.int32), colidx[lo + s0:lo + s1])
            assert torch.equal(_as_tensor(p["vals"] + 8 * s0, s1 - s0, "<f8", dev, torch.float64), vals[lo + s0:lo + s1] * 2.0)
        seen["rows"] += nr
        seen["nnz"] += p["nnz"]
    info = ctx.spgemm_csc_csr_panels(np.float64, n, n, n, [csc[0].data_ptr(), csc[1].data_ptr(), a2.data_ptr(), *ptrs[3:]], on_panel)
    note("streamed (2A)*B compared")
    assert seen["rows"] == n and seen["nnz"] == nnz == info["nnz_c"] and info["panels"] >= 2
    res.close()
    del rowptr, colidx, vals, a2
    ctx.trim()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("how", ["atomic", "ballot"])
def test_dense_accumulation_both_ways(port, monkeypatch, how):
    """The dense accumulators add either with one LDS floating-point atomic per 64 entries (lanes that hit one
    accumulator are applied in lane order -- self-tested for f32 and f64 when the context is created) or by ballot ranks
    and rounds (the fallback).  OSP_DENSE_ADD forces either: same bits as the oracle on products whose hub rows go through them,
    with exact cancellation and a lone negative zero among the piles."""
    from outerspace_amd import spgemm as S
    monkeypatch.setenv("OSP_DENSE_ADD", how)
    with S.Context(0) as c:
        for preset, scale, dt in (("g500", 14, np.float64), ("g500", 13, np.float32)):
            n, rows, cols, vals = gen.rmat_coo(scale, 16, preset, seed=9, dtype=dt)
            vals = (vals * np.where(np.arange(len(vals)) % 3 == 0, -1, 1)).astype(dt)   # mixed signs: order matters
            got, want = run_both(c, port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt)
            assert got.info["dense_segments"] > 0
            assert_same(got, want)
        rng = np.random.default_rng(3)
        K = 20000
        a = (np.zeros(K, np.uint32), np.arange(K, dtype=np.uint32), rng.uniform(-1.0, 1.0, K) * 10.0 ** rng.integers(-12, 12, K))
        bcols = rng.integers(0, 6, K).astype(np.uint32)
        bv = rng.uniform(-1.0, 1.0, K)
        bv[bcols == 4] = 0.0
        bv[np.flatnonzero(bcols == 4)[0]] = -0.0          # column 4: a pile of zeros that starts with a negative one
        first5 = np.flatnonzero(bcols == 5)
        bv[first5] = 0.0
        bv[first5[0]], bv[first5[1]] = 1.0, -a[2][first5[0]] / a[2][first5[1]]   # column 5 cancels exactly
        got, want = run_both(c, port, 2, K, 8, a, (np.arange(K, dtype=np.uint32), bcols, bv), np.float64)
        assert got.info["dense_segments"] > 0
        assert_same(got, want)
        assert np.signbit(got.vals[4]) == np.signbit(want["vals"][4])


def test_ballot_rank_fallback_is_exact(port, monkeypatch):
    """Every ranking kernel is compiled twice: stable ranks from the return order of one LDS atomic (the default, after a
    self-test of that undocumented property when the context is created) and from ballot matching (what a context falls
    back to when the self-test fails).  OSP_RANK=ballot forces the fallback: same bits as the oracle on inputs that reach
    every ranking kernel -- tile sort, both long-row splits, the global-sort path, the symbolic sort, the device ingest."""
    from outerspace_amd import spgemm as S
    monkeypatch.setenv("OSP_RANK", "ballot")
    with S.Context(0) as c:
        for algo in ("outer", "rowwise"):
            c.algorithm = algo
            for preset, scale, dt in (("g500", 13, np.float64), ("mild", 12, np.float32)):
                n, rows, cols, vals = gen.rmat_coo(scale, 16, preset, seed=5, dtype=dt)
                got, want = run_both(c, port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt)
                assert got.info["heavy_rows"] > 0
                assert_same(got, want)
        monkeypatch.setenv("OSP_SPLIT_ROW_MAX", "0")      # every long row through the stretch split
        monkeypatch.setenv("OSP_BIGTILE_CAP", "0")        # every over-long segment through the global sort
        monkeypatch.setenv("OSP_DENSE_SEG", "0")          # (not the dense accumulators: test_dense_accumulation_both_ways)
        n, rows, cols, vals = gen.rmat_coo(12, 16, "g500", seed=6)
        got, want = run_both(c, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float64)
        assert got.info["sorted_segments"] > 0
        assert_same(got, want)
        rng = np.random.default_rng(3)
        p = rng.permutation(len(rows))
        got = c.spgemm_coo(n, n, n, (rows[p], cols[p], vals[p]), (rows, cols, vals))
        assert_same(got, want)


@pytest.mark.parametrize("preset,scale,dt", [("mild", 19, np.float64), ("g500", 16, np.float64), ("mild", 17, np.float32)])
def test_panel_boundaries_sweep(ctx, preset, scale, dt):
    """Where the row panels are cut depends on the staging capacity (by default: on the free memory of the moment), and
    every cut moves the wave slices of the multiply, the tiles and the long-row splits.  A hundred capacities, resident and
    streamed: every one must give the bits of the one-panel product."""
    import torch
    from outerspace_amd.distributed import _as_tensor
    dev = torch.device("cuda", 0)
    n, csr, csc = _bench_module().rmat_device(scale, 16, gen.RMAT_PRESETS[preset], 5, dev, torch.float64 if dt == np.float64 else torch.float32)
    torch.cuda.synchronize()
    fdt, tdt, idt = ("<f8", torch.float64, torch.int64) if dt == np.float64 else ("<f4", torch.float32, torch.int32)
    ptrs = [t.data_ptr() for t in (*csc, *csr)]
    ref = ctx.spgemm_csc_csr_device(dt, n, n, n, ptrs)
    P, nnz = ref.info["partials"], ref.nnz
    assert ref.info["panels"] == 1
    rp, ci, va = ref.device_ptrs()
    rowptr, colidx, vals = (_as_tensor(rp, n + 1, "<i8", dev, torch.int64), _as_tensor(ci, nnz, "<i4", dev, torch.int32),
                            _as_tensor(va, nnz, fdt, dev, tdt))
    rng = np.random.default_rng(int(os.environ.get("OSP_SWEEP_SEED", "99")))
    for it in range(int(os.environ.get("OSP_SWEEP_ITERS", "100"))):   # (soak runs: OSP_SWEEP_ITERS=1000 OSP_SWEEP_SEED=...)
        cap = int(rng.integers(P // 14, P // 2))
        if it % 2 == 0:
            try:
                res = ctx.spgemm_csc_csr_device(dt, n, n, n, ptrs, partial_capacity=cap)
            except RuntimeError as e:   # a hub row alone can exceed a small capacity: that is an error by contract
                assert "staging capacity" in str(e)
                continue
            assert res.nnz == nnz and res.info["panels"] > 1, (cap, res.nnz, nnz)
            r2, c2, v2 = res.device_ptrs()
            assert torch.equal(_as_tensor(r2, n + 1, "<i8", dev, torch.int64), rowptr), cap
            assert torch.equal(_as_tensor(c2, nnz, "<i4", dev, torch.int32), colidx), cap
            assert torch.equal(_as_tensor(v2, nnz, fdt, dev, tdt).view(idt), vals.view(idt)), cap
            res.close()
        else:
            def on_panel(p):
                lo = int(rowptr[p["row_begin"]])
                nr = p["row_end"] - p["row_begin"]
                prp = _as_tensor(p["rowptr"], nr + 1, "<i8", dev, torch.int64)
                assert torch.equal(prp, rowptr[p["row_begin"]:p["row_end"] + 1] - lo), (cap, p["index"])
                assert p["nnz"] == int(rowptr[p["row_end"]]) - lo, (cap, p["index"])
                if p["nnz"]:
                    assert torch.equal(_as_tensor(p["colidx"], p["nnz"], "<i4", dev, torch.int32), colidx[lo:lo + p["nnz"]]), (cap, p["index"])
                    assert torch.equal(_as_tensor(p["vals"], p["nnz"], fdt, dev, tdt).view(idt),
                                       vals[lo:lo + p["nnz"]].view(idt)), (cap, p["index"])
            try:
                info = ctx.spgemm_csc_csr_panels(dt, n, n, n, ptrs, on_panel, partial_capacity=cap)
            except RuntimeError as e:
                assert "staging capacity" in str(e)
                continue
            assert info["nnz_c"] == nnz and info["panels"] > 1
    ref.close()


@pytest.mark.parametrize("preset,scale", [("mild", 18), ("g500", 15)])
def test_slab_and_shard_sweep(ctx, preset, scale):
    """The units the multi-GPU modes compute -- a k slab of the operands, a row shard of the result, both at once -- each at
    a random staging capacity against the same unit in one panel: forty random units, same bits."""
    import torch
    from outerspace_amd.distributed import _as_tensor
    dev = torch.device("cuda", 0)
    n, csr, csc = _bench_module().rmat_device(scale, 16, gen.RMAT_PRESETS[preset], 11, dev, torch.float64)
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in (*csc, *csr)]
    rng = np.random.default_rng(int(os.environ.get("OSP_SWEEP_SEED", "7")))

    def unit(**kw):
        r = ctx.spgemm_csc_csr_device(np.float64, n, n, n, ptrs, **kw)
        rp, ci, va = r.device_ptrs()
        nr = r.info["row_end"] - r.info["row_begin"] if kw.get("row_shard") else n
        out = (_as_tensor(rp, nr + 1, "<i8", dev, torch.int64).clone(), _as_tensor(ci, r.nnz, "<i4", dev, torch.int32).clone(),
               _as_tensor(va, r.nnz, "<f8", dev, torch.float64).view(torch.int64).clone(), r.info["partials"], r.info["panels"])
        torch.cuda.synchronize()   # the copies run on torch's stream: done before the buffers go back to the library's pool
        r.close()                  # (OSP_POISON=1 refills them at once, on the library's stream)
        return out
    done = 0
    for it in range(int(os.environ.get("OSP_SWEEP_ITERS", "40"))):
        kw = {}
        if it % 3 != 0:
            k0 = int(rng.integers(0, n - 1))
            kw["k_range"] = (k0, int(rng.integers(k0 + 1, min(n, k0 + 1 + n // int(rng.integers(1, 9))) + 1)))
        if it % 2 == 1:
            G = int(rng.integers(2, 6))
            kw["row_shard"] = (int(rng.integers(0, G)), G)
        ref = unit(**kw)
        if ref[3] < 4096:
            continue
        cap = int(rng.integers(ref[3] // 9 + 1, ref[3] // 2 + 2))
        try:
            got = unit(partial_capacity=cap, **kw)
        except RuntimeError as e:   # one row alone can exceed a small capacity: an error by contract
            assert "staging capacity" in str(e)
            continue
        assert got[3] == ref[3] and got[4] > 1, (kw, cap)
        for a, b in zip(got[:3], ref[:3]):
            assert torch.equal(a, b), (kw, cap)
        done += 1
    assert done >= 20


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_long_runs_inside_tiles(ctx, port, dt):
    """Runs longer than kRunShort inside ordinary tiles (an output entry fed by dozens to hundreds of products) are summed by
    a whole wave, 64 values per round trip, added one by one in staging order: the bits of the one-lane walk.  Rows that fit
    a tile (level 0), rows that are split (segment tiles), runs that end exactly at the 8th / 64th entry and at the tile's
    end."""
    rng = np.random.default_rng(23)
    M, K, N = 96, 700, 4096
    rows, cols = [], []
    for i in range(M):
        nk = 250 if i % 2 == 0 else 650   # ~900 products (one tile) or ~2300 (split into segments)
        ks = np.sort(rng.choice(K, nk, replace=False))
        rows.append(np.full(nk, i)); cols.append(ks)
    a_rows = np.concatenate(rows).astype(np.uint32); a_cols = np.concatenate(cols).astype(np.uint32)
    a = (a_rows, a_cols, rng.uniform(-1.0, 1.0, len(a_rows)).astype(dt))
    # every B row: column 7 always (runs of 250 / 650), one of 16 hub columns (runs of ~15 / ~40), two anywhere
    hub = 100 + 37 * rng.integers(0, 16, K)
    anyc = rng.integers(2000, N, (K, 2))
    cols = np.concatenate([np.full((K, 1), 7), hub[:, None], anyc], 1)
    cols.sort(axis=1)
    keep = np.ones(cols.shape, bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    b_rows = np.repeat(np.arange(K, dtype=np.uint32), 4)[keep.reshape(-1)]
    b_cols = cols.reshape(-1)[keep.reshape(-1)].astype(np.uint32)
    b = (b_rows, b_cols, rng.uniform(-1.0, 1.0, len(b_cols)).astype(dt))
    got, want = run_both(ctx, port, M, K, N, a, b, dt)
    assert_same(got, want)
    # runs of exactly 8, 9, 64, 65 and 72 products, and one that fills a whole tile's worth of a row
    for L in (8, 9, 64, 65, 72, 1500):
        a1 = (np.zeros(L, np.uint32), np.arange(L, dtype=np.uint32), rng.uniform(-1.0, 1.0, L).astype(dt))
        b1 = (np.arange(L, dtype=np.uint32), np.full(L, 3, np.uint32), rng.uniform(-1.0, 1.0, L).astype(dt))
        got, want = run_both(ctx, port, 2, L, 16, a1, b1, dt)
        assert_same(got, want)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_hub_segments_dense_accumulation(ctx, port, monkeypatch, dt):
    """Over-long segments with a narrow column range -- hub row x hub columns, the Graph500 case -- are reduced by one LDS
    accumulator per column, entries taken in staging order and equal columns of one 64-entry group added in rank order:
    the bits of the sort-and-sum.  Cases: a row whose products pile up on a handful of columns (thousands of duplicates per
    column), exact cancellation inside a pile (the zero stays an entry), a lone negative zero, and Graph500 skew."""
    rng = np.random.default_rng(17)
    M, K, N = 4, 9000, 512
    # row 0 of A is dense in k; every B row has three entries among 12 hub columns + one elsewhere
    a = (np.zeros(K, np.uint32), np.arange(K, dtype=np.uint32), rng.uniform(0.5, 1.5, K).astype(dt))
    hub = rng.integers(0, 12, (K, 3))
    hub.sort(axis=1)
    cols = np.concatenate([hub, rng.integers(12, N, (K, 1))], 1)
    keep = np.ones(cols.shape, bool)
    keep[:, 1:3] = cols[:, 1:3] != cols[:, 0:2]
    b_rows = np.repeat(np.arange(K, dtype=np.uint32), 4)[keep.reshape(-1)]
    b_cols = cols.reshape(-1)[keep.reshape(-1)].astype(np.uint32)
    b_vals = rng.uniform(-1.0, 1.0, len(b_cols)).astype(dt)
    # exact cancellation on column 3: the entries of the first two B rows that hit it cancel, nothing else hits it
    sel3 = b_cols == 3
    b_vals[sel3] = 0.0
    first = np.flatnonzero(sel3)[:2]
    a_k = a[2][b_rows[first]]
    b_vals[first[0]] = 1.0 / a_k[0].astype(np.float64) if False else dt(1.0)
    b_vals[first[1]] = dt(-(a_k[0] * dt(1.0)) / a_k[1])
    got, want = run_both(ctx, port, M, K, N, a, (b_rows, b_cols, b_vals), dt)
    assert got.info["dense_segments"] >= 1 and got.info["sorted_segments"] == 0
    assert_same(got, want)
    # a lone negative zero in a pile of zeros keeps its sign only if the sum starts AS the first entry
    b2 = (np.arange(K, dtype=np.uint32), np.zeros(K, np.uint32), np.zeros(K, dt))
    b2[2][0] = dt(-0.0)
    got, want = run_both(ctx, port, M, K, N, (a[0], a[1], np.ones(K, dt)), b2, dt)
    assert_same(got, want)
    assert np.signbit(got.vals[0]) == np.signbit(want["vals"][0])
    # Graph500 skew at a size the oracle still handles
    n, rows, cols_, vals = gen.rmat_coo(14, 16, "g500", seed=2, dtype=dt)
    got, want = run_both(ctx, port, n, n, n, (rows, cols_, vals), (rows, cols_, vals), dt)
    assert_same(got, want)


@pytest.mark.parametrize("hub", ["1", "0"])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_hub_rows_written_by_the_multiply(port, monkeypatch, _ctx_shared, hub, dt):
    """Rows beyond the one-workgroup planner (here: beyond OSP_SPLIT_ROW_MAX products) are hub rows: the multiply writes them
    into uniform column blocks from one cell per (chunk, run of B's row) -- hub_plan_kernel -- instead of the stretch split
    moving them afterwards (OSP_HUB=0).  Graph500 skew (hub row x hub column: long runs), a row fed by thousands of one-entry
    chunks (runs of length one), empty chunks in between, several panels, a k range: all bit for bit against the oracle."""
    from outerspace_amd import spgemm as S
    monkeypatch.setenv("OSP_HUB", hub)
    monkeypatch.setenv("OSP_HUB_MIN_SHARE", "0")   # (by default only panels whose products are mostly in such rows plan them this way ...
    monkeypatch.setenv("OSP_HUB_MIN_RUN", "0")     #  ... and only where a run holds four records on average)
    monkeypatch.setenv("OSP_SPLIT_ROW_MAX", "3000")
    monkeypatch.setenv("OSP_DIRECT_MAX", "3000")
    c = _ctx_shared
    c.algorithm = "outer"
    n, rows, cols, vals = gen.rmat_coo(13, 16, "g500", seed=5, dtype=dt)
    # B: the same pattern with its columns spread over 2^20 (with few columns every long row is a "dense" direct row instead)
    N = n * 128
    bcols = (cols.astype(np.uint64) * 128 + (rows.astype(np.uint64) * 31 + cols.astype(np.uint64) * 17) % 128).astype(np.uint32)
    for kw in ({}, {"partial_capacity": 300000}, {"k_range": (100, 5000)}):
        got, want = run_both(c, port, n, n, N, (rows, cols, vals), (rows, bcols, vals), dt, **kw)
        assert_same(got, want)
        if hub == "1":
            assert got.info["hub_rows"] > 0 and got.info["hub_partials"] > 0, got.info
        else:
            assert got.info["hub_rows"] == 0
    # one output row fed by 40 000 chunks of one or two entries each (and some of no entry at all), spread over all columns
    rng = np.random.default_rng(23)
    M, K, N = 3, 40000, 70000
    a = (np.concatenate([np.zeros(K, np.uint32), np.full(50, 2, np.uint32)]), np.concatenate([np.arange(K, dtype=np.uint32), np.arange(50, dtype=np.uint32) * 7]),
         rng.uniform(0.5, 1.5, K + 50).astype(dt))
    nb = rng.integers(0, 3, K)
    b_rows = np.repeat(np.arange(K, dtype=np.uint32), nb)
    b_cols = rng.integers(0, N, len(b_rows)).astype(np.uint32)
    keep = np.ones(len(b_rows), bool)
    keep[1:] = (b_rows[1:] != b_rows[:-1]) | (b_cols[1:] != b_cols[:-1])
    order = np.lexsort((b_cols, b_rows))
    b_rows, b_cols = b_rows[order], b_cols[order]
    keep = np.ones(len(b_rows), bool)
    keep[1:] = (b_rows[1:] != b_rows[:-1]) | (b_cols[1:] != b_cols[:-1])
    b_rows, b_cols = b_rows[keep], b_cols[keep]
    b = (b_rows, b_cols, rng.uniform(-1, 1, len(b_rows)).astype(dt))
    got, want = run_both(c, port, M, K, N, a, b, dt)
    assert_same(got, want)
    if hub == "1":
        assert got.info["hub_rows"] == 1
        # the same row with the default threshold: its runs are single records, the panel keeps the stretch split (same bits)
        monkeypatch.setenv("OSP_HUB_MIN_RUN", "4")
        got, want = run_both(c, port, M, K, N, a, b, dt)
        assert_same(got, want)
        assert got.info["hub_rows"] == 0


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_next_panels_plan_runs_beside_the_multiply(port, monkeypatch, _ctx_shared, overlap):
    """A product of several panels plans panel p+1 on the context's second stream while panel p is multiplied
    (osp_pipeline.h, merge_pipeline; OSP_PLAN_OVERLAP=0 plans in line).  Direct rows, hub rows and split rows in one product,
    resident and streamed, twice in a row on one context (the pool hands the first product's buffers to the second): the
    same bits as the oracle either way, and the info says how many plans ran beside a multiply."""
    from outerspace_amd import spgemm as S
    monkeypatch.setenv("OSP_PLAN_OVERLAP", overlap)
    monkeypatch.setenv("OSP_HUB_MIN_SHARE", "0")
    monkeypatch.setenv("OSP_HUB_MIN_RUN", "0")
    monkeypatch.setenv("OSP_SPLIT_ROW_MAX", "3000")
    monkeypatch.setenv("OSP_DIRECT_MAX", "2000")
    c = _ctx_shared
    c.algorithm = "outer"
    for dt, preset, scale in ((np.float64, "g500", 13), (np.float32, "mild", 14)):
        n, rows, cols, vals = gen.rmat_coo(scale, 16, preset, seed=9, dtype=dt)
        N = n * 64
        bcols = (cols.astype(np.uint64) * 64 + (rows.astype(np.uint64) * 13 + cols.astype(np.uint64) * 7) % 64).astype(np.uint32)
        for rep in range(2):
            got, want = run_both(c, port, n, n, N, (rows, cols, vals), (rows, bcols, vals), dt, partial_capacity=200000)
            assert_same(got, want)
            assert got.info["panels"] > 3
            assert got.info["plans_overlapped"] == (got.info["panels"] - 1 if overlap == "1" else 0), got.info
            got.close()
        # streamed: the panels as they are finished
        import torch
        from outerspace_amd.distributed import _as_tensor
        dev = torch.device("cuda:0")
        acsc, bcsr = S.coo_to_csc(n, rows, cols, vals.astype(dt)), S.coo_to_csr(n, rows, bcols, vals.astype(dt))
        t = [torch.from_numpy(a.astype(np.int32) if a.dtype == np.uint32 else a).to(dev) for a in (*acsc, *bcsr)]
        tdt, fmt = (torch.float64, "<f8") if dt == np.float64 else (torch.float32, "<f4")
        parts = []

        def on_panel(p):
            parts.append((_as_tensor(p["colidx"], p["nnz"], "<i4", dev, torch.int32).cpu().numpy().view(np.uint32),
                          _as_tensor(p["vals"], p["nnz"], fmt, dev, tdt).cpu().numpy()))
        info = c.spgemm_csc_csr_panels(dt, n, n, N, [x.data_ptr() for x in t], on_panel, partial_capacity=200000)
        assert info["panels"] == len(parts) > 3
        assert info["plans_overlapped"] == (info["panels"] - 1 if overlap == "1" else 0)
        assert np.array_equal(np.concatenate([q[0] for q in parts]), want["colidx"])
        assert np.array_equal(np.concatenate([q[1] for q in parts]), want["vals"])


def test_context_on_the_callers_stream(port, monkeypatch):
    """osp_context_create_on_stream: the product's work is ordered on the caller's stream -- operands written by kernels
    queued on that stream just before the call, the result read by a kernel queued on it right after, no synchronisation in
    between -- although a product of several panels plans on a second stream of the context's own (forked from and joined
    into the caller's by events)."""
    import torch
    from outerspace_amd import spgemm as S
    from outerspace_amd.distributed import _as_tensor
    monkeypatch.setenv("OSP_DIRECT_MIN_NNZ", "0")
    monkeypatch.setenv("OSP_DIRECT_MAX", "3000")
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(device=dev)
    c = S.Context(0, stream=st.cuda_stream)
    try:
        n, rows, cols, vals = gen.rmat_coo(14, 16, "mild", seed=6)
        acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
        want = port.spgemm(n, n, n, acsc[0], acsc[1], acsc[2] * 2.0, bcsr[0], bcsr[1], bcsr[2] * 0.5)
        host = [torch.from_numpy(a.astype(np.int32) if a.dtype == np.uint32 else a).pin_memory() for a in (*acsc, *bcsr)]
        with torch.cuda.stream(st):
            t = [h.to(dev, non_blocking=True) for h in host]
            t[2] = t[2] * 2.0      # A's and B's values are final only when these kernels, queued on the stream, have run
            t[5] = t[5] * 0.5
            res = c.spgemm_csc_csr_device(np.float64, n, n, n, [x.data_ptr() for x in t], partial_capacity=want["partials"] // 5 + 1)
            _, _, va = res.device_ptrs()
            total = _as_tensor(va, res.nnz, "<f8", dev, torch.float64).sum()
        assert res.info["panels"] > 3 and res.info["plans_overlapped"] == res.info["panels"] - 1 and res.info["direct_rows"] > 0
        assert float(total) == pytest.approx(float(np.sum(want["vals"])), rel=1e-12)
        assert_same(res, want)
        res.close()
    finally:
        c.close()


@pytest.mark.parametrize("direct_max", [None, "0", "5000", "40000"])
def test_direct_rows_written_by_the_multiply(port, monkeypatch, _ctx_shared, direct_max):
    """Long rows that one workgroup could split are written straight into their column ranges by the multiply phase
    (osp_split.h, direct_plan_kernel): no split pass over their records.  OSP_DIRECT_MAX bounds the rows it applies to, so
    direct rows, split rows and stretch-split rows meet in one product; the panel variant cuts the product into several
    row panels (every panel plans its own direct rows).  Same bits as the oracle every way."""
    if direct_max is not None:
        monkeypatch.setenv("OSP_DIRECT_MAX", direct_max)
    monkeypatch.setenv("OSP_SPLIT_ROW_MAX", "60000")     # rows beyond: the stretch split
    ctx = _ctx_shared
    for dt, preset, scale in ((np.float64, "mild", 15), (np.float32, "g500", 13)):
        n, rows, cols, vals = gen.rmat_coo(scale, 16, preset, seed=4, dtype=dt)
        for cap in (0, 1 << 21):
            got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt, partial_capacity=cap)
            assert got.info["heavy_rows"] > 50
            if direct_max == "0":
                assert got.info["direct_rows"] == 0 and got.info["split_launches"] >= 1
            else:
                assert got.info["direct_rows"] > 0 and got.info["direct_partials"] > 0 and got.info["direct_plan_launches"] >= 1
                assert got.info["direct_rows"] <= got.info["heavy_rows"]
            if cap:
                assert got.info["panels"] > 1
            assert_same(got, want)
            got.close()


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_dense_long_rows_are_direct_rows(port, monkeypatch, _ctx_shared, dt):
    """Rows with more products than one workgroup splits (here: more than OSP_SPLIT_ROW_MAX = 4096) whose column ranges were
    capped at the width of a dense accumulator -- few columns, many products per column -- are written by range by the
    multiply phase as well and summed by the dense accumulators (osp_split.h, split_params_kernel: `capped`): no stretch split.
    n = 2048 columns (two ranges of 1024 columns), ~100 entries per row, ~10 000 products per output row."""
    monkeypatch.setenv("OSP_SPLIT_ROW_MAX", "4096")
    ctx = _ctx_shared
    n, rows, cols, vals = gen.rmat_coo(11, 100, "uniform", seed=9, dtype=dt)
    for cap in (0, 1 << 22):
        got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt, partial_capacity=cap)
        i = got.info
        assert i["heavy_rows"] == n and i["direct_rows"] == n and i["direct_partials"] == i["heavy_partials"] > 4096 * n
        assert i["split_launches"] == 0
        if cap:
            assert i["panels"] > 1
        assert_same(got, want)
        got.close()


@pytest.mark.parametrize("scale,ef,preset,row_max,direct_max,cap", [
    (10, 64, "uniform", "2048", None, 0),          # every row long and dense: two ranges, all direct
    (10, 64, "uniform", "2048", "0", 1 << 20),     # the same rows through the stretch split, several panels
    (12, 100, "uniform", "8192", "16384", 0),      # capped rows beyond OSP_DIRECT_MAX: still direct (dense), by the cap
    (12, 48, "g500", "4096", None, 1 << 21),       # skew: hub rows far beyond 2^20 / row_max products next to short ones
    (13, 32, "mild", "1024", "3000", 0),           # narrow limits: direct, split and stretch rows in one product
    (9, 200, "uniform", "60000", None, 0),         # 512 columns: a single range per row
])
def test_dense_and_small_column_counts(port, monkeypatch, _ctx_shared, scale, ef, preset, row_max, direct_max, cap):
    """Matrices with few columns (2^9 .. 2^13) whose long rows hold many products per column: the column ranges of such
    rows are capped at the dense accumulators' width (osp_pipeline.h, bits_cap) and the rows are written by range directly
    whatever their length (osp_split.h, `capped`).  Limits moved so that every combination of direct / split / stretch rows and
    of dense / sorted segments occurs at sizes the oracle forms in seconds; same bits as the oracle."""
    monkeypatch.setenv("OSP_SPLIT_ROW_MAX", row_max)
    if direct_max is not None:
        monkeypatch.setenv("OSP_DIRECT_MAX", direct_max)
    ctx = _ctx_shared
    for dt in (np.float64, np.float32):
        n, rows, cols, vals = gen.rmat_coo(scale, ef, preset, seed=scale + ef, dtype=dt)
        got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt, partial_capacity=cap)
        assert got.info["heavy_rows"] > 0
        if cap:
            assert got.info["panels"] > 1
        assert_same(got, want)
        got.close()


def test_direct_rows_by_density_for_small_operands(port, monkeypatch, _ctx_shared):
    """A product of few non-zeros keeps the split (OSP_DIRECT_MIN_NNZ, here: out of reach) -- unless its output rows are dense
    on average (>= 0.375 partial products per entry of the M x N result): then its long rows are direct rows all the same."""
    monkeypatch.setenv("OSP_DIRECT_MIN_NNZ", "1000000000")
    ctx = _ctx_shared
    n, rows, cols, vals = gen.rmat_coo(10, 64, "uniform", seed=3)      # ~3.7 M partial products for a 1024 x 1024 result
    got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float64)
    assert got.info["partials"] * 8 >= 3 * n * n and got.info["heavy_rows"] > 0 and got.info["direct_rows"] == got.info["heavy_rows"]
    assert_same(got, want)
    got.close()
    n, rows, cols, vals = gen.rmat_coo(13, 16, "g500", seed=3)         # skewed, sparse result: long rows, but no density
    got, want = run_both(ctx, port, n, n, n, (rows, cols, vals), (rows, cols, vals), np.float64)
    assert got.info["partials"] * 8 < 3 * n * n and got.info["heavy_rows"] > 0 and got.info["direct_rows"] == 0
    assert_same(got, want)
    got.close()


def test_record_parts_are_validated(_ctx_shared):
    """osp_merge_record_parts with cfg.validate: a column beyond N or a broken offset array is an error code, not an
    out-of-bounds access on the device."""
    import torch
    from outerspace_amd import _lib
    from outerspace_amd import spgemm as S
    ctx = _ctx_shared
    dev = torch.device("cuda", 0)
    M, N = 4, 10
    rec = np.zeros(6, dtype=S.aos_dtype(np.float64))
    rec["idx"] = [1, 3, 9, 0, 2, 7]
    rec["val"] = np.arange(6) + 1.0
    rp = np.array([0, 2, 3, 3, 6], np.int64)

    def run(rp_h, rec_h, validate):
        t_rp = torch.from_numpy(rp_h).to(dev)
        t_rec = torch.from_numpy(rec_h.view(np.uint8)).to(dev)
        torch.cuda.synchronize()
        return ctx.merge_record_parts_device(np.float64, M, N, [(t_rp.data_ptr(), t_rec.data_ptr())], validate=validate)

    ok = run(rp, rec, True)
    assert ok.nnz == 6 and np.array_equal(ok.colidx, [1, 3, 9, 0, 2, 7])
    ok.close()
    bad = rec.copy()
    bad["idx"][2] = 10
    with pytest.raises(S.OspError) as ei:
        run(rp, bad, True)
    assert ei.value.status == _lib.ERR_RANGE
    with pytest.raises(S.OspError) as ei:
        run(np.array([0, 3, 2, 3, 6], np.int64), rec, True)
    assert ei.value.status == _lib.ERR_ARG


def test_two_devices_one_thread():
    """Contexts on two devices owned by one thread: the pinned staging object (and its events) is per device (ADVICE
    round 2: one thread-local object recorded device-0 events on a device-1 stream).  Needs two GPUs; the builder's box
    has one, the driver's 8-GPU node runs it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    from outerspace_amd import spgemm as S
    n, rows, cols, vals = gen.rmat_coo(9, 8, "mild", seed=3)
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    with S.Context(0) as c0, S.Context(1) as c1:
        r0 = c0.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
        r1 = c1.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
        r0b = c0.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
        for r in (r1, r0b):
            assert np.array_equal(r.rowptr, r0.rowptr) and np.array_equal(r.colidx, r0.colidx) and np.array_equal(r.vals, r0.vals)


@pytest.mark.parametrize("K", [3000, 60000])
def test_long_row_of_tiny_chunks(port, _ctx_shared, K):
    """One output row fed by K chunks of one or two entries (a dense row of A times a near-diagonal B): many (chunk, range)
    cells per product.  Below the planner's per-row cell budget the row is written directly (K = 3000), above it the row is
    split after the multiply (K = 60000) -- same bits either way, no pathological planning time."""
    rng = np.random.default_rng(11)
    M, N = 3, 50000
    a = (np.zeros(K, np.uint32), np.arange(K, dtype=np.uint32), rng.uniform(0.5, 1.5, K))
    b_rows = np.concatenate([np.arange(K), np.arange(0, K, 3)]).astype(np.uint32)
    b_cols = np.concatenate([(np.arange(K) * 7919) % N, (np.arange(0, K, 3) * 104729 + 1) % N]).astype(np.uint32)
    key = np.unique(b_rows.astype(np.int64) * N + b_cols)
    b = ((key // N).astype(np.uint32), (key % N).astype(np.uint32), rng.uniform(0.5, 1.5, len(key)))
    got, want = run_both(_ctx_shared, port, M, K, N, a, b, np.float64)
    assert got.info["heavy_rows"] == 1
    assert got.info["direct_rows"] == (1 if K == 3000 else 0)
    assert_same(got, want)
    got.close()


@pytest.mark.parametrize("switch", ["OSP_GATHER=1", "OSP_GATHER_OVER=0", "OSP_EXPAND_ROWS=0", "default"])
def test_gathered_rows_partial_settings(_ctx_shared, port, monkeypatch, switch):
    """Gathered rows with parts of the scheme switched off, so that the paths they replace stay alive beside
    them in ONE product: planned long rows gathered but short rows staged by the compacted column-major multiply
    (OSP_GATHER=1); rows with a range that exceeds a tile written by the multiply through cells while the others are gathered
    (OSP_GATHER_OVER=0); rows beyond the planner staged column by column from the compacted list instead of row by row
    (OSP_EXPAND_ROWS=0).  Skewed inputs with small planner limits: hub rows, over-long segments, rows of many blocks of
    chunks, several panels.  All bit-identical to the oracle; the info says which rows were gathered."""
    if switch != "default":
        k, v = switch.split("=")
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("OSP_SPLIT_ROW_MAX", "20000")
    monkeypatch.setenv("OSP_DIRECT_MAX", "12000")
    c = _ctx_shared
    c.algorithm = "outer"
    for dt, preset, scale, cap in ((np.float64, "g500", 13, 0), (np.float32, "g500", 12, 60000), (np.float64, "mild", 14, 300000)):
        n, rows, cols, vals = gen.rmat_coo(scale, 16, preset, seed=11, dtype=dt)
        got, want = run_both(c, port, n, n, n, (rows, cols, vals), (rows, cols, vals), dt, partial_capacity=cap)
        assert_same(got, want)
        i = got.info
        assert i["direct_rows"] > 0 and i["gathered_rows"] > 0 and i["gathered_runs"] > 0, i
        if switch == "OSP_GATHER_OVER=0" and preset == "g500":
            assert i["gathered_rows"] < i["direct_rows"], i     # the rows with an over-long range were written
        if switch in ("default", "OSP_EXPAND_ROWS=0", "OSP_GATHER_OVER=0"):
            assert i["gathered_short_partials"] > 0, i
        if switch == "OSP_GATHER=1":
            assert i["gathered_short_partials"] == 0, i
        if switch == "default":
            assert i["gathered_rows"] == i["direct_rows"], i
        got.close()
