"""Seeded synthetic operands for tests and bench (host side, numpy).

No reference counterpart: the reference ships no generator or input data for the SpGEMM path
(SURVEY.md section 6).  Duplicate coordinates are removed because the reference's ``coo2csr``
rejects them (``simulator/SimSpGEMM.cpp:43-53,123``).
"""
import numpy as np

RMAT_PRESETS = {
    "uniform": (0.25, 0.25, 0.25, 0.25),
    "mild": (0.45, 0.22, 0.22, 0.11),
    "g500": (0.57, 0.19, 0.19, 0.05),
}


def rmat_coo(scale, edge_factor=16, abcd="g500", seed=1, dtype=np.float64):
    """R-MAT edge list on n = 2**scale vertices, duplicates removed, sorted by (row, col).

    Returns (n, rows u32, cols u32, vals dtype) with vals ~ U(0.5, 1.5).
    """
    a, b, c, d = RMAT_PRESETS[abcd] if isinstance(abcd, str) else abcd
    n = 1 << scale
    m = edge_factor * n
    rng = np.random.default_rng(seed)
    rows = np.zeros(m, np.int64)
    cols = np.zeros(m, np.int64)
    for _ in range(scale):
        u = rng.random(m)
        # quadrants in order a (0,0), b (0,1), c (1,0), d (1,1)
        rbit = u >= a + b
        cbit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
        rows = (rows << 1) | rbit
        cols = (cols << 1) | cbit
    key = np.unique(rows * n + cols)
    rows = (key // n).astype(np.uint32)
    cols = (key % n).astype(np.uint32)
    vals = rng.uniform(0.5, 1.5, len(key)).astype(dtype)
    return n, rows, cols, vals


def random_coo(nrow, ncol, density, seed=0, dtype=np.float64):
    """Uniform random pattern without duplicates, values U(0,1); sorted by (row, col)."""
    rng = np.random.default_rng(seed)
    nnz = int(round(density * nrow * ncol))
    key = np.sort(rng.choice(nrow * ncol, size=nnz, replace=False))
    rows = (key // ncol).astype(np.uint32)
    cols = (key % ncol).astype(np.uint32)
    vals = rng.random(nnz).astype(dtype)
    return rows, cols, vals


def coo_to_csr(nrow, rows, cols, vals):
    """COO (any order, no duplicates) -> CSR arrays (rowptr i64, colidx u32, vals)."""
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = np.zeros(nrow + 1, np.int64)
    np.add.at(rowptr, rows.astype(np.int64) + 1, 1)
    return np.cumsum(rowptr), np.ascontiguousarray(cols, np.uint32), np.ascontiguousarray(vals)


def coo_to_csc(ncol, rows, cols, vals):
    """COO -> CSC arrays (colptr i64, rowidx u32, vals)."""
    return coo_to_csr(ncol, cols, rows, vals)


def mlp_layer_operands(H, threshold):
    """BASELINE configs[4] at its stated shape, rebuilt from a seed: act_0 (1024 x 784, ~16 % dense, like a post-ReLU
    activation) and fc1_weight (H x 784) pruned to |w| > threshold -- the inputs tests/golden/make_golden.py:case_mlp_full
    fed to the compiled reference (the threshold it stored came from the reference's get_prune_threshold).
    Returns (act f32 dense, pruned W f32 dense, COO of act, COO of W^T)."""
    rng = np.random.default_rng(1024 + H)
    W = (rng.standard_normal((H, 784)) * 0.05).astype(np.float32)
    act = np.maximum(rng.standard_normal((1024, 784)).astype(np.float32) - np.float32(1.0), np.float32(0.0))
    Wp = W * (np.abs(W) > np.float32(threshold))
    ar, ac = np.nonzero(act)
    wr, wc = np.nonzero(Wp)
    a = (ar.astype(np.uint32), ac.astype(np.uint32), act[ar, ac])
    b = (wc.astype(np.uint32), wr.astype(np.uint32), Wp[wr, wc])   # B = W^T: (k, out)
    return act, W, Wp, a, b
