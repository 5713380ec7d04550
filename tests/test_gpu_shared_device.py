"""Regression test for round 1's intermittent GPU memory fault.

The persistent merge kernel hands out tiles by ticket.  A workgroup that starts late -- because the CUs are busy with
another kernel or another process -- can find every ticket taken; it then held a tile descriptor slot in LDS that was
never filled, and indexing the kernel-argument arrays with that garbage faulted.  With the GPU to itself the library
hit this perhaps once in sixty small, skewed products; beside a second process that keeps the CUs busy it failed every
time.  So this test runs products whose tile counts are close to the merge grid while a neighbour process
(tools/gpu_neighbor.py, plain torch matmuls) shares the device, and checks every result against the oracle."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from outerspace_amd import generators as gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_products_beside_a_busy_neighbour_process(ctx, port):
    from outerspace_amd import spgemm as S
    nb = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gpu_neighbor.py"), "compute", "22"],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        time.sleep(8)  # its start-up (import torch, first matmul) -- from here on the CUs are contended
        assert nb.poll() is None, nb.stdout.read().decode(errors="replace")
        n, rows, cols, vals = gen.rmat_coo(14, 16, "g500", seed=5)
        acsc = S.coo_to_csc(n, rows, cols, vals)
        bcsr = S.coo_to_csr(n, rows, cols, vals)
        cuts = [0, n // 64, n // 8, n // 2, n]
        wants = [port.spgemm(n, n, n, *acsc, *bcsr, cuts[i], cuts[i + 1]) for i in range(4)]
        t_end = time.time() + 8
        rounds = 0
        while time.time() < t_end:
            for i in range(4):  # k slabs of very different skew: few tiles to many, over-long segments in the first
                got = ctx.spgemm_csc_csr(n, n, n, *acsc, *bcsr, k_range=(cuts[i], cuts[i + 1]), validate=False)
                assert np.array_equal(got.rowptr, wants[i]["rowptr"]) and np.array_equal(got.colidx, wants[i]["colidx"])
                assert np.array_equal(got.vals, wants[i]["vals"])
                got.close()
            rounds += 1
        assert rounds >= 1
        assert nb.poll() is None, "the neighbour ended early: " + nb.stdout.read().decode(errors="replace")
    finally:
        nb.kill()
        nb.wait()
