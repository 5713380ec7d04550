#!/usr/bin/env python3
"""Planning aid (CPU, numpy): how long would the runs be if the multiply wrote every product of a long output row
straight into its column range?  For an R-MAT self-product: rows by partial-product count U_i, and -- on a sample of
long rows -- the non-empty (chunk, range) pairs when the row is cut into ranges of <= CAP products.
    python tools/run_stats.py [scale] [abcd] [samples]
"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from outerspace_amd.generators import rmat_coo, coo_to_csr

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
abcd = sys.argv[2] if len(sys.argv) > 2 else "mild"
nsamp = int(sys.argv[3]) if len(sys.argv) > 3 else 400
CAP = 1536
n, r, c, v = rmat_coo(scale, 16, abcd, 1)
rowptr, colidx, _ = coo_to_csr(n, r, c, v)
del v
nb = np.diff(rowptr)                       # nnz(B[k,:]), B = A
U = np.zeros(n, np.int64)
np.add.at(U, r.astype(np.int64), nb[c.astype(np.int64)])   # U_i = sum_k nb_k over A[i,:]
P = int(U.sum())
deg = nb
print(f"n={n} nnz={len(r)} P={P:.4g}")
edges = [0, CAP, 4096, 16384, 65536, 131072, 1 << 20, 1 << 40]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (U > lo) & (U <= hi)
    T = np.ceil(U[m] / 1400.0)
    cells = float((deg[m] * T).sum())
    print(f"U in ({lo},{hi}]: rows {int(m.sum())}, products {U[m].sum()/P*100:.1f} %, chunks {int(deg[m].sum())}, "
          f"avg chunk {U[m].sum()/max(1,deg[m].sum()):.1f}, dense cells/product {cells/max(1,U[m].sum()):.3f}")
rng = np.random.default_rng(0)
for lo, hi in zip(edges[1:-1], edges[2:]):
    rows = np.nonzero((U > lo) & (U <= hi))[0]
    if len(rows) == 0:
        continue
    # sample rows with probability proportional to U (we want per-product statistics)
    pick = rng.choice(rows, size=min(nsamp, len(rows)), p=U[rows] / U[rows].sum())
    tot = runs_t = runs_f = 0
    lens = []
    for i in pick:
        ks = colidx[rowptr[i]:rowptr[i + 1]].astype(np.int64)
        segs = [colidx[rowptr[k]:rowptr[k + 1]] for k in ks]
        cols = np.concatenate(segs).astype(np.int64)
        chunk = np.repeat(np.arange(len(ks)), [len(s) for s in segs])
        Ui = len(cols)
        # fine bins of ~256 products: 2^b uniform column ranges (as split_params_kernel does)
        want = -(-Ui // 256)
        b = 1
        while b < 12 and (1 << b) < want:
            b += 1
        sh = max(0, scale - b)
        fine = cols >> sh
        hist = np.bincount(fine, minlength=1 << b)
        # greedy grouping of fine bins into ranges of <= CAP
        tile = np.zeros(1 << b, np.int64)
        t = 0
        acc = 0
        for d in range(1 << b):
            if acc + hist[d] > CAP and acc > 0:
                t += 1
                acc = 0
            tile[d] = t
            acc += hist[d]
        tl = tile[fine]
        runs_t += len(np.unique(chunk * (t + 1) + tl))
        runs_f += len(np.unique(chunk * (1 << b) + fine))
        tot += Ui
    print(f"  sample U in ({lo},{hi}]: products/run at range level {tot/runs_t:.1f}, at fine-bin level {tot/runs_f:.1f}")
