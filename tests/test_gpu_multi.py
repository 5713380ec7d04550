"""The k-sharded product INSIDE the library (osp_multi_*, include/outerspace_spgemm.h): G ranks driven by one call --
slabs of k on their ranks, partial products copied to the owner of their row behind the multiply, one merge per row range.
The builder's box has one GPU, so the ranks here are logical ranks sharing device 0: the same code path (contexts,
streams, events, peer copies -- from device 0 to device 0 --, rank threads, barriers), bit for bit against the oracle.
On a box with several GPUs the same test spreads the ranks over them."""
import os

import numpy as np
import pytest

from outerspace_amd import generators as gen

pytestmark = pytest.mark.gpu


def _devices(nranks):
    import torch
    nd = max(1, torch.cuda.device_count())
    return [g % nd for g in range(nranks)]


@pytest.mark.parametrize("nranks", [1, 2, 3, 4])
@pytest.mark.parametrize("preset,scale,dt", [("mild", 12, np.float64), ("g500", 11, np.float32)])
def test_multi_gpu_product_is_bit_identical(port, nranks, preset, scale, dt):
    from outerspace_amd import spgemm as S
    n, rows, cols, vals = gen.rmat_coo(scale, 12, preset, seed=31, dtype=dt)
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    want = port.spgemm(n, n, n, *acsc, *bcsr)
    with S.MultiGpu(_devices(nranks)) as mg:
        info, (rowptr, colidx, v) = mg.spgemm_csc_csr(n, n, n, *acsc, *bcsr, validate=True)
        assert info["nranks"] == nranks and info["partials"] == want["partials"] and info["nnz_c"] == len(want["colidx"])
        assert np.array_equal(rowptr, want["rowptr"]) and np.array_equal(colidx, want["colidx"])
        assert np.array_equal(v, want["vals"])     # parts merged in rank order = ascending k: the single-GPU bits
        ranks = info["ranks"]
        assert ranks[0]["k_begin"] == 0 and ranks[-1]["k_end"] == n and ranks[0]["row_begin"] == 0 and ranks[-1]["row_end"] == n
        assert all(a["k_end"] == b["k_begin"] and a["row_end"] == b["row_begin"] for a, b in zip(ranks, ranks[1:]))
        assert sum(r["partials_local"] for r in ranks) == want["partials"] == sum(r["records_received"] for r in ranks)
        assert sum(r["nnz_c"] for r in ranks) == info["nnz_c"]
        if nranks > 1:
            assert info["bytes_exchanged"] > 0 and all(r["bytes_sent"] > 0 for r in ranks)
            # one copy stream per destination; what went where adds up, nothing is "sent" to oneself
            for g, r in enumerate(ranks):
                assert r["copy_streams"] == nranks - 1 and sum(r["bytes_to"]) == r["bytes_sent"] and r["bytes_to"][g] == 0
                assert r["max_copies_in_flight"] >= 1 and r["max_copies_outstanding"] >= 1 and r["ms_exchange"] > 0
            # what rank g sent to rank h is what h received from others, in total
            assert sum(r["bytes_sent"] for r in ranks) == info["bytes_exchanged"]
            # slabs balanced by partial products
            assert max(r["partials_local"] for r in ranks) < 2.5 * want["partials"] / nranks
        # the loaded slabs stay resident: a second product gives the same result
        info2, (rowptr2, colidx2, v2) = mg.multiply()
        assert np.array_equal(rowptr2, rowptr) and np.array_equal(colidx2, colidx) and np.array_equal(v2, v)


def test_multi_gpu_exchange_fans_out(monkeypatch):
    """The exchange of osp_multi.h: every destination has a copy stream and send slots of its own, so the G-1 pieces of a
    sub-panel are all queued -- each waiting only for its own piece to be multiplied -- before the host looks again.  On one
    GPU (logical ranks) the copies are device-to-device and as fast as the multiply, so whether their DMA intervals overlap
    is only asserted where the ranks sit on different devices.  Result against the one-GPU product, bit for bit."""
    import torch
    from outerspace_amd import spgemm as S
    monkeypatch.setenv("OSP_MULTI_SUBPANELS", "4")
    n, rows, cols, vals = gen.rmat_coo(16, 16, "mild", seed=7)
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    with S.Context(0) as ctx:
        one = ctx.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
        want = (one.rowptr.copy(), one.colidx.copy(), one.vals.copy())
        one.close()
    G = 4
    with S.MultiGpu(_devices(G)) as mg:
        info, (rowptr, colidx, v) = mg.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
        assert np.array_equal(rowptr, want[0]) and np.array_equal(colidx, want[1]) and np.array_equal(v, want[2])
        assert info["subpanels"] == 4
        for g, r in enumerate(info["ranks"]):
            assert r["copy_streams"] == G - 1
            assert all(b > 0 for h, b in enumerate(r["bytes_to"]) if h != g)
            # (how many copies the host SAW queued at once, or overlapped on the DMA engines, depends on host / device timing
            # when the ranks share one GPU: reported, not asserted, there; structure is what must hold)
            assert r["max_copies_outstanding"] >= 1, info["ranks"]
            if torch.cuda.device_count() >= G:
                assert r["max_copies_in_flight"] >= 2, info["ranks"]


@pytest.mark.parametrize("subpanels", ["1", "7"])
def test_multi_gpu_pipeline_granularity_and_panels(port, monkeypatch, subpanels):
    """Sub-panels per row range (the unit of the multiply / copy / merge pipeline) and a small staging capacity (the merge
    cuts its panels further): same bits.  Rectangular operands, empty rows and an empty slab included."""
    from outerspace_amd import spgemm as S
    monkeypatch.setenv("OSP_MULTI_SUBPANELS", subpanels)
    rng = np.random.default_rng(5)
    M, K, N = 700, 300, 900
    a = gen.random_coo(M, K, 0.03, seed=1)
    b = gen.random_coo(K, N, 0.04, seed=2)
    keep = a[0] % 7 != 3            # some empty rows
    a = tuple(x[keep] for x in a)
    acsc, bcsr = S.coo_to_csc(K, *a), S.coo_to_csr(K, *b)
    want = port.spgemm(M, K, N, *acsc, *bcsr)
    with S.MultiGpu(_devices(3)) as mg:
        for cap in (0, 4096):
            info, (rowptr, colidx, v) = mg.spgemm_csc_csr(M, K, N, *acsc, *bcsr, partial_capacity=cap)
            assert np.array_equal(rowptr, want["rowptr"]) and np.array_equal(colidx, want["colidx"]) and np.array_equal(v, want["vals"])
    # all of k in one column: two of three slabs are empty
    a1 = (np.arange(50, dtype=np.uint32), np.full(50, 4, np.uint32), rng.uniform(1, 2, 50))
    b1 = (np.full(60, 4, np.uint32), np.arange(60, dtype=np.uint32), rng.uniform(1, 2, 60))
    acsc, bcsr = S.coo_to_csc(8, *a1), S.coo_to_csr(8, *b1)
    want = port.spgemm(50, 8, 60, *acsc, *bcsr)
    with S.MultiGpu(_devices(3)) as mg:
        info, (rowptr, colidx, v) = mg.spgemm_csc_csr(50, 8, 60, *acsc, *bcsr)
        assert np.array_equal(rowptr, want["rowptr"]) and np.array_equal(colidx, want["colidx"]) and np.array_equal(v, want["vals"])


def test_multi_gpu_errors(port):
    from outerspace_amd import _lib
    from outerspace_amd import spgemm as S
    n, rows, cols, vals = gen.rmat_coo(8, 8, "mild", seed=3)
    acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
    with pytest.raises(S.OspError):
        S.MultiGpu([])
    with pytest.raises(S.OspError) as ei:
        S.MultiGpu([0, 99])
    assert ei.value.status == _lib.ERR_ARG
    with S.MultiGpu([0, 0]) as mg:
        bad = acsc[1].copy()
        bad[1] = bad[0]      # a duplicate row index inside column 0 (if it has two entries) or an unsorted pair
        if acsc[0][1] >= 2:
            with pytest.raises(S.OspError) as ei:
                mg.spgemm_csc_csr(n, n, n, acsc[0], bad, acsc[2], *bcsr, validate=True)
            assert ei.value.status in (_lib.ERR_DUPLICATE, _lib.ERR_UNSORTED)
        # and the context still works afterwards
        info, out = mg.spgemm_csc_csr(n, n, n, *acsc, *bcsr)
        want = port.spgemm(n, n, n, *acsc, *bcsr)
        assert np.array_equal(out[1], want["colidx"]) and np.array_equal(out[2], want["vals"])


def test_cli_gpus_flag(golden_dir, tmp_path):
    """`osp_spgemm A.mtx B.mtx --gpus 3` (the CLI keeps the reference's two positional paths and gains
    --gpus): same header lines, same result file as the one-GPU run, 233 on a duplicate coordinate."""
    import subprocess
    from outerspace_amd import spgemm as S
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "outerspace_amd", "osp_spgemm")
    a, b = os.path.join(golden_dir, "c1_A.mtx"), os.path.join(golden_dir, "c1_B.mtx")
    outs = []
    for extra in ([], ["--gpus", "3"]):
        out = tmp_path / f"c{len(extra)}.mtx"
        r = subprocess.run([exe, a, b, "--f64", "--out", str(out)] + extra, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        assert "NCol = 64, NRow = 64, NNZ = 410" in r.stdout and "mul flops ref = 2692" in r.stdout
        outs.append(S.read_mtx(str(out)))
        if extra:
            assert "rank 2 (device" in r.stdout and "GPU x3" in r.stdout
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x, y)
    dup = tmp_path / "dup.mtx"
    dup.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n2 2 2.0\n2 2 3.0\n")
    r = subprocess.run([exe, str(dup), str(dup), "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 233
