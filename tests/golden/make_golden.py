#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Run in the build container only (needs /root/reference):

    make -C oracle && python tests/golden/make_golden.py

Sources of truth used here
  * oracle/_ref/libref_f{32,64}.so -- the reference's simulator/SimSpGEMM.cpp compiled in place
    (readcoo :55-100, coo2csr :102-152, cscMulcsr :265-281, sort+sum :519-535).
  * NN_models/util.py:61-62 save_tensor_as_mtx and NN_models/sparse_util.py:5-22 imported from
    /root/reference (Python side), for the .mtx byte format and the pruning helpers.

Only DATA is written: input matrices and the reference's outputs.  No reference source text.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from outerspace_amd import generators as gen  # noqa: E402

REF_NN = "/root/reference/NN_models"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def write_mtx_plain(path, nrow, ncol, rows, cols, vals, fmt="%.17g"):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%\n")
        f.write(f"{nrow} {ncol} {len(rows)}\n")
        for r, c, v in zip(rows, cols, vals):
            f.write(f"{r + 1} {c + 1} {fmt % v}\n")


def case_c1():
    """BASELINE.json configs[0]: 64x64, 10 % dense, A * B^T through .mtx files."""
    import scipy.sparse as sp
    import torch
    sys.path.insert(0, REF_NN)
    import util as ref_util  # reference: NN_models/util.py

    rng = np.random.default_rng(0)
    A = sp.random(64, 64, 0.10, random_state=rng, dtype=np.float32, format="csr")
    B = sp.random(64, 64, 0.10, random_state=rng, dtype=np.float32, format="csr")
    pa, pb = os.path.join(HERE, "c1_A.mtx"), os.path.join(HERE, "c1_B.mtx")
    ref_util.save_tensor_as_mtx(torch.from_numpy(A.toarray()), pa)
    ref_util.save_tensor_as_mtx(torch.from_numpy(B.toarray()), pb)
    out = {}
    for dt in (np.float32, np.float64):
        rc, M, N, P, r, c, v = oracle.ref(dt).spgemm_mtx(pa, pb, transpose_b=True)
        assert rc == 0
        s = np.dtype(dt).name
        out.update({f"rows_{s}": r, f"cols_{s}": c, f"vals_{s}": v})
        out.update(M=M, N=N, P=P)
        # and without the transpose workaround (A*B), f64 only
    rc, M, N, P, r, c, v = oracle.ref(np.float64).spgemm_mtx(pa, pb, transpose_b=False)
    out.update(nt_rows=r, nt_cols=c, nt_vals=v, nt_P=P)
    np.savez_compressed(os.path.join(HERE, "c1_expected.npz"), **out)
    print("c1: P =", out["P"], "nnzC =", len(out["rows_float64"]))


def case_reader():
    """readcoo quirks: banner/comments/blank lines/tabs/pattern entries/exponent floats."""
    path = os.path.join(HERE, "reader_quirks.mtx")
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("% a comment\n")
        f.write("\n")
        f.write("   \t \n")
        f.write("  % indented comment\n")
        f.write("4 5 7\n")
        f.write("1 1 1.5\n")
        f.write("\n")
        f.write("2\t3\t-2.25e-1\n")
        f.write("4 5\n")              # pattern entry -> 1.0
        f.write("% mid comment\n")
        f.write("3 2 1e3\n")
        f.write("  1 5   0.333333343\n")
        f.write("2 1 7 trailing junk\n")
        f.write("4 1 0.1\n")
    out = {}
    for dt in (np.float32, np.float64):
        nrow, ncol, r, c, v = oracle.ref(dt).readcoo(path)
        s = np.dtype(dt).name
        out.update({f"rows_{s}": r, f"cols_{s}": c, f"vals_{s}": v, "nrow": nrow, "ncol": ncol})
    np.savez_compressed(os.path.join(HERE, "reader_quirks_expected.npz"), **out)
    print("reader: nnz =", len(out["rows_float64"]))


def _run_csx(M, K, N, a, b, dt):
    """a, b: COO triples of A (MxK) and B (KxN).  Uses the REFERENCE coo2csr and multiply/merge."""
    ref = oracle.ref(dt)
    rc, apos, aidx, aval = ref.coo2csr(True, K, a[0], a[1], a[2].astype(dt))
    assert rc == 0, rc
    rc, bpos, bidx, bval = ref.coo2csr(False, K, b[0], b[1], b[2].astype(dt))
    assert rc == 0, rc
    res = ref.spgemm_csx(K, apos, aidx, aval, bpos, bidx, bval)
    return dict(apos=apos, aidx=aidx, aval=aval, bpos=bpos, bidx=bidx, bval=bval, **res)


def case_edges():
    out = {}
    # (1) rectangular 5x7 * 7x3 with an empty row of A (row 2), an empty column of A (k=4),
    #     and a k whose B row is empty (k=1) although A[:,1] is not.
    a = (np.array([0, 0, 1, 3, 3, 4, 4, 1], np.uint32), np.array([0, 1, 2, 0, 6, 5, 2, 3], np.uint32),
         np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]))
    b = (np.array([0, 0, 2, 2, 3, 4, 5, 6, 6], np.uint32), np.array([0, 2, 1, 2, 0, 1, 2, 0, 1], np.uint32),
         np.array([0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5]))
    for dt in (np.float32, np.float64):
        r = _run_csx(5, 7, 3, a, b, dt)
        s = np.dtype(dt).name
        out.update({f"rect_{k}_{s}": v for k, v in r.items() if k != "secs"})
    out.update(rect_a_rows=a[0], rect_a_cols=a[1], rect_a_vals=a[2],
               rect_b_rows=b[0], rect_b_cols=b[1], rect_b_vals=b[2])

    # (2) exact cancellation: C[0,0] = 1*2 + 2*(-1) = 0 must stay as an explicit zero.
    a = (np.array([0, 0, 1], np.uint32), np.array([0, 1, 1], np.uint32), np.array([1.0, 2.0, 3.0]))
    b = (np.array([0, 1, 1], np.uint32), np.array([0, 0, 1], np.uint32), np.array([2.0, -1.0, 4.0]))
    r = _run_csx(2, 2, 2, a, b, np.float64)
    out.update({f"cancel_{k}": v for k, v in r.items() if k != "secs"})
    out.update(cancel_a_rows=a[0], cancel_a_cols=a[1], cancel_a_vals=a[2],
               cancel_b_rows=b[0], cancel_b_cols=b[1], cancel_b_vals=b[2])

    # (3) duplicate coordinate -> the reference throws 233.
    rows = np.array([0, 1, 1, 2], np.uint32)
    cols = np.array([0, 2, 2, 1], np.uint32)
    vals = np.array([1.0, 2.0, 3.0, 4.0])
    rc_csr, *_ = oracle.ref(np.float64).coo2csr(False, 3, rows, cols, vals)
    rc_csc, *_ = oracle.ref(np.float64).coo2csr(True, 3, rows, cols, vals)
    out.update(dup_rows=rows, dup_cols=cols, dup_vals=vals, dup_rc_csr=rc_csr, dup_rc_csc=rc_csc)

    # (4) the back-fill quirk (SimSpGEMM.cpp:143-148): one non-empty segment -> reference
    #     returns pos[] all == nnz (an EMPTY matrix).  Recorded to document the divergence.
    rows = np.array([2, 2, 2], np.uint32)
    cols = np.array([0, 1, 3], np.uint32)
    vals = np.array([1.0, 2.0, 3.0])
    rc, pos, idx, val = oracle.ref(np.float64).coo2csr(False, 4, rows, cols, vals)
    out.update(onerow_rows=rows, onerow_cols=cols, onerow_vals=vals, onerow_rc=rc, onerow_ref_pos=pos)

    # (5) unsorted input with trailing empty segments, both orientations.
    rows = np.array([3, 0, 1, 3, 0, 1], np.uint32)
    cols = np.array([1, 4, 0, 0, 2, 5], np.uint32)
    vals = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    for tr, nseg in ((0, 6), (1, 8)):
        rc, pos, idx, val = oracle.ref(np.float64).coo2csr(bool(tr), nseg, rows, cols, vals)
        out.update({f"conv{tr}_rc": rc, f"conv{tr}_pos": pos, f"conv{tr}_idx": idx, f"conv{tr}_val": val})
    out.update(conv_rows=rows, conv_cols=cols, conv_vals=vals)
    np.savez_compressed(os.path.join(HERE, "edges_expected.npz"), **out)
    print("edges: rect nnzC =", out["rect_nnzc_float64"], "cancel nnzC =", out["cancel_nnzc"],
          "dup rc =", rc_csr, rc_csc, "onerow pos =", out["onerow_ref_pos"])


def case_rmat():
    """Skewed self-product (G500 R-MAT, scale 10): digests only."""
    out = {}
    for dt in (np.float32, np.float64):
        n, rows, cols, vals = gen.rmat_coo(10, 16, "g500", seed=1, dtype=dt)
        r = _run_csx(n, n, n, (rows, cols, vals), (rows, cols, vals), dt)
        rowptr, _, _ = oracle.coo_to_csr(n, r["rows"], r["cols"], r["vals"])
        s = np.dtype(dt).name
        out.update({
            f"nnzA_{s}": len(rows), f"P_{s}": r["partials"], f"nnzC_{s}": r["nnzc"],
            f"in_sha_{s}": sha(rows) + sha(cols) + sha(vals),
            f"rowptr_sha_{s}": sha(rowptr.astype(np.int64)), f"colidx_sha_{s}": sha(r["cols"]),
            f"val_sum_{s}": np.float64(r["vals"].astype(np.float64).sum()),
            f"val_abs_max_{s}": np.float64(np.abs(r["vals"]).max()),
            # a sparse sample of values for a tolerance check
            f"sample_idx_{s}": np.arange(0, r["nnzc"], 997, dtype=np.int64),
            f"sample_val_{s}": r["vals"][::997].copy(),
        })
    np.savez_compressed(os.path.join(HERE, "rmat10_expected.npz"), **out)
    print("rmat10: nnzA =", out["nnzA_float64"], "P =", out["P_float64"], "nnzC =", out["nnzC_float64"])


def case_mlp():
    """BASELINE.json configs[4] shape, reduced batch: act(64x784) * W(100x784)^T, f32.

    Weights are pruned with the reference's threshold (sparse_util.py:9-10) in the |w| > thr form
    main.py:208-211 uses; both matrices are written by the reference's save_tensor_as_mtx.
    """
    import torch
    sys.path.insert(0, REF_NN)
    import sparse_util as ref_su
    import util as ref_util

    torch.manual_seed(0)
    W = torch.randn(100, 784) * 0.05
    thr = ref_su.get_prune_threshold(W, 0.05)
    Wp = W * (W.abs() > thr)
    act = torch.relu(torch.randn(64, 784) - 1.0)  # ~16 % dense, like a post-ReLU activation
    pa, pw = os.path.join(HERE, "mlp_act.mtx"), os.path.join(HERE, "mlp_fc1_weight.mtx")
    ref_util.save_tensor_as_mtx(act, pa)
    ref_util.save_tensor_as_mtx(Wp, pw)
    rc, M, N, P, r, c, v = oracle.ref(np.float32).spgemm_mtx(pa, pw, transpose_b=True)
    assert rc == 0
    dense = (act.double() @ Wp.double().T).numpy()
    cnt, numel, frac = ref_su.get_sparsity(Wp)
    np.savez_compressed(os.path.join(HERE, "mlp_expected.npz"), M=M, N=N, P=P, rows=r, cols=c, vals=v,
                        dense_f64=dense, prune_thr=np.float32(thr), w_nnz=int(cnt), w_numel=int(numel),
                        w_frac=np.float32(frac))
    # tensor for the sparse_util pins
    np.savez_compressed(os.path.join(HERE, "mlp_weight_dense.npz"), W=W.numpy(), Wp=Wp.numpy())
    print("mlp: P =", P, "nnzC =", len(r), "thr =", float(thr), "w nnz =", int(cnt))


def mlp_full_inputs(H, thr=None):
    """configs[4] at its stated shape: act_0 (1024 x 784, the reference's batch: config.py:2, get_mtx_files.py:19-73) and
    fc1_weight (H x 784, models.py:10-14; H = 100, or 1000 in the saved logs), both from numpy's seeded generator so that
    the tests can rebuild them without the reference.  Returns (act f32, W f32 dense, sparsity level)."""
    rng = np.random.default_rng(1024 + H)
    W = (rng.standard_normal((H, 784)) * 0.05).astype(np.float32)
    act = np.maximum(rng.standard_normal((1024, 784)).astype(np.float32) - np.float32(1.0), np.float32(0.0))
    level = 0.05 if H == 100 else 0.01   # "pruned to 1 % of the weights": saved_weights/MLP1/prune0p01_l2reg/log.txt
    return act, W, level


def case_mlp_full():
    """BASELINE.json configs[4] at full shape (batch 1024, H in {100, 1000}): weights pruned with the REFERENCE's threshold
    (sparse_util.py:9-10, in main.py:208-211's |w| > thr form), product act * W^T from the compiled reference in f32.
    Stored: the threshold, digests of the structure, a sample of the values and the f64 dense values at the sample."""
    import torch
    sys.path.insert(0, REF_NN)
    import sparse_util as ref_su
    out = {}
    for H in (100, 1000):
        act, W, level = mlp_full_inputs(H)
        thr = np.float32(ref_su.get_prune_threshold(torch.from_numpy(W), level))
        Wp = W * (np.abs(W) > thr)
        cnt, numel, frac = ref_su.get_sparsity(torch.from_numpy(Wp))
        ar, ac = np.nonzero(act)
        wr, wc = np.nonzero(Wp)
        a = (ar.astype(np.uint32), ac.astype(np.uint32), act[ar, ac])
        b = (wc.astype(np.uint32), wr.astype(np.uint32), Wp[wr, wc])          # B = W^T: (k, out)
        r = _run_csx(1024, 784, H, a, b, np.float32)
        rowptr, _, _ = oracle.coo_to_csr(1024, r["rows"], r["cols"], r["vals"])
        idx = np.arange(0, r["nnzc"], 997, dtype=np.int64)
        dense = (act.astype(np.float64) @ Wp.astype(np.float64).T)[r["rows"][idx], r["cols"][idx]]
        out.update({f"thr_{H}": thr, f"level_{H}": np.float32(level), f"w_nnz_{H}": int(cnt), f"act_nnz_{H}": len(ar),
                    f"in_sha_{H}": sha(act) + sha(W), f"P_{H}": r["partials"], f"nnzC_{H}": r["nnzc"],
                    f"rowptr_sha_{H}": sha(rowptr.astype(np.int64)), f"colidx_sha_{H}": sha(r["cols"]),
                    f"val_sum_{H}": np.float64(r["vals"].astype(np.float64).sum()), f"val_abs_max_{H}": np.float32(np.abs(r["vals"]).max()),
                    f"sample_idx_{H}": idx, f"sample_val_{H}": r["vals"][idx].copy(), f"sample_dense_f64_{H}": dense})
        print(f"mlp full H={H}: act nnz {len(ar)}, W nnz {int(cnt)} ({float(frac):.4f}), thr {float(thr):.6f}, P = {r['partials']}, nnzC = {r['nnzc']}")
    np.savez_compressed(os.path.join(HERE, "mlp_full_expected.npz"), **out)


if __name__ == "__main__":
    assert oracle.have_ref(), "build oracle/_ref first: make -C oracle"
    case_c1()
    case_reader()
    case_edges()
    case_rmat()
    case_mlp()
    case_mlp_full()
