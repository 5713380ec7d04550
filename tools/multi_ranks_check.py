"""Eight (then five) logical ranks on ONE GPU through osp_multi_*: every rank thread, copy stream and event of the exchange at once,
three products each, against the one-GPU result bit for bit.  A robustness check beside tests/test_gpu_multi.py (1-4 ranks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from outerspace_amd import generators as gen
from outerspace_amd import spgemm as S
n, rows, cols, vals = gen.rmat_coo(17, 16, "mild", seed=11)
acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
with S.Context(0) as ctx:
    one = ctx.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
    want = (one.rowptr.copy(), one.colidx.copy(), one.vals.copy())
    one.close()
for G in (8, 5):
    with S.MultiGpu([0] * G) as mg:
        mg.load(n, n, n, *acsc, *bcsr)
        for rep in range(3):
            t0 = time.time()
            info, (rp, ci, v) = mg.multiply()
            ok = np.array_equal(rp, want[0]) and np.array_equal(ci, want[1]) and np.array_equal(v, want[2])
            print(G, "ranks, rep", rep, "identical" if ok else "DIFFERENT", round((time.time() - t0) * 1e3, 1), "ms (with fetch)",
                  "outstanding", [r["max_copies_outstanding"] for r in info["ranks"]], "in flight", [r["max_copies_in_flight"] for r in info["ranks"]])
            assert ok
print("ok")
