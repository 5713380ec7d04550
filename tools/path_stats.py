#!/usr/bin/env python3
"""Planning aid (CPU): which multiply path the partial products of an R-MAT self-product take (by length of B's row)."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from outerspace_amd.generators import rmat_coo
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
abcd = sys.argv[2] if len(sys.argv) > 2 else "mild"
n, r, c, v = rmat_coo(scale, 16, abcd, 1)
nb = np.bincount(r, minlength=n).astype(np.int64)   # row counts of B = A
na = np.bincount(c, minlength=n).astype(np.int64)   # column counts of A
prod = na * nb
P = prod.sum()
for lo, hi in ((0, 32), (32, 64), (64, 1024), (1024, 1 << 30)):
    m = (nb > lo) & (nb <= hi)
    print(f"nb in ({lo},{hi}]: columns {int(m.sum())}, products {prod[m].sum() / P * 100:.1f} %")
