"""Two rank PROCESSES on one MI355X through the real multi-GPU code: `spgemm_k_sharded` (slab operands only -> local
product on the GPU -> all-to-all-v of the partial CSRs -> osp_merge_csr_parts on the GPU) and `spgemm_row_sharded`, with
the exchange staged through host memory over `gloo` (RCCL needs one GPU per rank; the driver's 8-GPU run uses it).
Every rank's rows of C are compared with the plain-C oracle: indices exact; values bit-exact for the row shards and for
the k shards that exchange their partial products unmerged (the default), within re-association of the slab sums (1e-12)
for the k shards that merge locally first."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, preset, scale, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from outerspace_amd import distributed as D
    from outerspace_amd import generators as gen
    from outerspace_amd import spgemm as S
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        n, rows, cols, vals = gen.rmat_coo(scale, 12, preset, seed=21)
        acsc, bcsr = S.coo_to_csc(n, rows, cols, vals), S.coo_to_csr(n, rows, cols, vals)
        to_dev = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)
        csc, csr = tuple(to_dev(a) for a in acsc), tuple(to_dev(a) for a in bcsr)
        D.A2A_MAX_BYTES = 1 << 16   # several exchange rounds, as at scale 22 where a pairwise message exceeds the RCCL limit
        with S.Context(0) as ctx:
            kb = D.plan_k_shards(csc[0], csr[0], world)
            slab = D.slice_k_slab(csc, csr, kb[rank], kb[rank + 1])
            assert slab[0] == kb[rank + 1] - kb[rank] and slab[1][0].numel() == slab[0] + 1 and int(slab[1][0][0]) == 0
            del csc, csr   # from here on the rank holds nothing but its slab
            # "f": rank 1's records "do not fit" after the ranks agreed on the raw exchange -- both must end up merged
            for tag, exch, fail in (("k", "raw", None), ("m", "merged", None), ("f", "raw", 1)):
                info = D.spgemm_k_sharded(ctx, np.float64, n, n, slab, dist, rank, world, stage_through_host=True, checksum=True,
                                          fetch=True, exchange=exch, _fail_raw_on_rank=fail)
                assert info["exchange"] == ("merged" if fail is not None else exch)
                rb = info["row_bounds"]
                np.savez(os.path.join(out_dir, f"{tag}{rank}.npz"), r0=rb[rank], r1=rb[rank + 1], rowptr=info["final_csr"][0],
                         colidx=info["final_csr"][1], vals=info["final_csr"][2], nnz_global=info["nnz_c_global"],
                         partials_global=info["partials_global"], val_sum=info["val_sum_global"], bytes_sent=info["bytes_sent"],
                         local_partials=info["partials"])
            # row shards: operands replicated, no exchange
            csc, csr = tuple(to_dev(a) for a in acsc), tuple(to_dev(a) for a in bcsr)
            info = D.spgemm_row_sharded(ctx, np.float64, n, n, n, [t.data_ptr() for t in (*csc, *csr)], dist, rank, world, dev,
                                        host_collectives=True, checksum=True, fetch=True)
            np.savez(os.path.join(out_dir, f"r{rank}.npz"), r0=info["row_begin"], r1=info["row_end"], rowptr=info["final_csr"][0],
                     colidx=info["final_csr"][1], vals=info["final_csr"][2], nnz_global=info["nnz_c_global"],
                     partials_global=info["partials_global"], val_sum=info["val_sum_global"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("preset,scale", [("mild", 12), ("g500", 11)])
def test_two_rank_processes_share_one_gpu(tmp_path, port, preset, scale):
    import torch.multiprocessing as mp
    from outerspace_amd import generators as gen
    from outerspace_amd import spgemm as S
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), preset, scale, str(tmp_path)), nprocs=world, join=True)
    n, rows, cols, vals = gen.rmat_coo(scale, 12, preset, seed=21)
    want = port.spgemm(n, n, n, *S.coo_to_csc(n, rows, cols, vals), *S.coo_to_csr(n, rows, cols, vals))
    for mode in ("k", "m", "f", "r"):
        got = [np.load(tmp_path / f"{mode}{r}.npz") for r in range(world)]
        assert int(got[0]["r0"]) == 0 and int(got[0]["r1"]) == int(got[1]["r0"]) and int(got[1]["r1"]) == n
        for g in got:
            r0, r1 = int(g["r0"]), int(g["r1"])
            lo, hi = want["rowptr"][r0], want["rowptr"][r1]
            assert np.array_equal(g["rowptr"], want["rowptr"][r0:r1 + 1] - lo)
            assert np.array_equal(g["colidx"], want["colidx"][lo:hi])
            if mode in ("r", "k"):
                # row shards, and k shards that send their partial products unmerged: one merge, in ascending k -- the bits
                # of the one-GPU product
                assert np.array_equal(g["vals"], want["vals"][lo:hi])
            else:
                assert np.allclose(g["vals"], want["vals"][lo:hi], rtol=1e-12, atol=0)     # slab sums re-associate
            assert int(g["nnz_global"]) == len(want["colidx"]) and int(g["partials_global"]) == want["partials"]
            assert np.isclose(float(g["val_sum"]), want["vals"].sum(), rtol=1e-12)
        if mode in ("k", "m", "f"):
            # each rank computed only its slab (about half the partial products) and sent about half of its partial CSR
            assert sum(int(g["local_partials"]) for g in got) == want["partials"]
            assert all(0 < int(g["bytes_sent"]) for g in got)
