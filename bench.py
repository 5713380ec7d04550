#!/usr/bin/env python3
"""Headline benchmark: output nnz/s of C = A*A for an R-MAT matrix (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W            # one MI355X
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W            # N GPUs, one rank each (the driver's launch)
    python bench.py --gpus N ...                             # same: spawns the N rank processes itself

A "step" is one complete product with the operands already resident in HBM: symbolic chunk layout, multiply, merge
into CSR.  For N > 1 one invocation measures the decompositions, each over W warm-up and K timed steps:
  k          the shared dimension k is cut into N slabs of equal partial-product count; a rank receives ONLY its columns of A
             and rows of B, multiplies, one all-to-all-v over RCCL exchanges the partial products by output row range, and each
             rank merges the pieces of its range (BASELINE.json north_star / configs[3]);
  k_library  the same decomposition inside the library (osp_spgemm_multi: one process drives the N GPUs, copies on one stream
             per destination behind the multiply, merge overlapped with the exchange), in a child process of rank 0 -- the
             line's `value` when it ran and its whole-result check passed, else `k`;
  rows       output rows sharded over the ranks, operands replicated, no exchange (SURVEY.md 8e's fallback for products whose
             exchange dominates).
All of them stay under "decompositions"; "fabric" says which backend, world size and devices the ranks saw.
Every run ends with an untimed whole-result check (1^T C 1 = (1^T A)(B 1)).  At N = 1: the CPU reference is timed on a
k-slab of the same matrix and the GPU's result for that slab is compared with it (indices exact, values <= 1e-6
relative); a plain copy is timed on the library's stream first (roofline.peak_measured: both fractions, of the data
sheet's 8 TB/s and of that copy); the default run adds five secondary workloads (extra_workloads), two of them --
the web-Google shape and the uniform R-MAT-22 -- with the CPU reference run IN FULL beside them and the whole GPU
result compared with it.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md
_T0 = time.time()


def note(what):
    """progress on stderr (rank 0): a run of several minutes must not look hung, and a crash should say where it was"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:6.1f} s] {what}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--rmat", default="mild",
                    help="R-MAT (a,b,c,d): mild=(.45,.22,.22,.11) [default: skewed, and C still fits one GPU's HBM] | "
                         "uniform=(.25,.25,.25,.25) | g500=(.57,.19,.19,.05) [scale-22 needs ~840 GB for C] | a,b,c,d")
    ap.add_argument("--workload", default="rmat", choices=["rmat", "webgoogle", "cage15"],
                    help="webgoogle = BASELINE configs[1]: the real SuiteSparse file when --a-mtx names it (or $OSP_WEBGOOGLE_MTX / "
                         "./web-Google.mtx exist), else its shape (916428 vertices, ~5.1 M pattern non-zeros, power-law degrees)")
    ap.add_argument("--a-mtx", default=None, help="MatrixMarket file of A: the product of real files instead of a synthetic matrix "
                                                  "(the reference CLI's call shape, SimSpGEMM.cpp:819-825)")
    ap.add_argument("--b-mtx", default=None, help="MatrixMarket file of B (default: the file of A)")
    ap.add_argument("--no-transpose-b", action="store_true",
                    help="with --a-mtx: multiply A * B; default A * B^T, as the reference CLI does (SimSpGEMM.cpp:852-856)")
    ap.add_argument("--ingest", type=int, default=1,
                    help="N=1: report the ingest beside the product -- host parse of the .mtx text (a bounded sample file for "
                         "synthetic workloads) and COO -> CSC/CSR on the device, next to the reference's Read Matrix / COO2CSR timers")
    ap.add_argument("--ingest-sample", type=int, default=4_000_000, help="entries of the sample file the parsers are timed on")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"])
    ap.add_argument("--partial-capacity", type=int, default=0)
    ap.add_argument("--algorithm", default="outer", choices=["outer", "rowwise"],
                    help="outer (default, the metric's algorithm) or the row-wise variant for rows that fit one merge tile")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="time the CPU reference on a k-slab (rank 0, N=1) and check the GPU against it")
    ap.add_argument("--cpu-partials", type=float, default=2.5e8, help="partial products in the CPU sample slab")
    ap.add_argument("--extras", type=int, default=1,
                    help="N=1, default workload only: also measure three secondary workloads (3 steps each) and report them "
                         "under 'extra_workloads': Graph500 parameters at scale 20 (streamed), uniform R-MAT-22, the web-Google shape")
    ap.add_argument("--stream-output", action="store_true",
                    help="single GPU: hand every finished row panel to a consumer (checksum) and drop it; C is never resident")
    ap.add_argument("--shard", default="both", choices=["both", "k", "rows"],
                    help="multi-GPU decomposition(s) to measure: k = shard the shared dimension, exchange partial CSRs over RCCL, "
                         "merge (the headline); rows = every rank computes a range of output rows from the replicated operands")
    ap.add_argument("--k-exchange", default="raw", choices=["raw", "merged"],
                    help="k-sharded product: what a rank sends -- raw = its partial products unmerged (one merge in all, bit-identical "
                         "to one GPU; default) | merged = its partial CSR (less to send when the product compresses well)")
    ap.add_argument("--library-multi-only", type=int, default=0, help="(child mode of --library-multi) ranks of the library's own product")
    ap.add_argument("--library-multi-timeout", type=int, default=600, help="seconds the child of --library-multi may take")
    ap.add_argument("--force-dist", type=int, default=0, help="run the distributed code paths even with one rank (sanity check)")
    ap.add_argument("--library-multi", type=int, default=1,
                    help="N>1: rank 0 also measures the library's own multi-GPU product (osp_spgemm_multi: ONE process driving all N "
                         "GPUs, partial products copied GPU to GPU behind the multiply, merge overlapped with the exchange) and "
                         "reports it under decompositions.k_library; the other ranks wait")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal only: exchange staged through host memory, ranks may share a GPU")
    return ap.parse_args()


# ---- launching N ranks from a plain `python bench.py --gpus N` ------------------------------------------------------------
def spawn_ranks(args):
    """No torchrun around us: start one process per GPU ourselves.  Nothing in THIS process has touched the GPU (no HIP
    call, no torch.cuda call), the children are fresh interpreters, and none of them replaces itself with another program."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OSP_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, set(range(len(procs)))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code and not rc:  # first failure: the others would wait in a collective for ever
                rc = code
                for q in live:
                    procs[q].terminate()
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


# ---- synthetic operands on the device ---------------------------------------------------------------------------------
def _compress(n, rows, cols, vals, device):
    """sorted-unique (rows, cols) -> CSR and CSC arrays (int64 ptr, int32 idx) of the same matrix"""
    import torch
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    colptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    colptr[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0)
    perm = torch.argsort(cols * n + rows)
    csr = (rowptr, cols.to(torch.int32).contiguous(), vals)
    csc = (colptr, rows[perm].to(torch.int32).contiguous(), vals[perm].contiguous())
    return csr, csc


def rmat_device(scale, ef, abcd, seed, device, dtype):
    """R-MAT on the GPU (same recipe as outerspace_amd.generators.rmat_coo), duplicates removed.
    Returns CSR and CSC arrays of the same matrix as torch tensors (int64 ptr, int32 idx)."""
    import torch
    n, m = 1 << scale, ef << scale
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if abcd == "regular":  # experiment only: every row has exactly `ef` non-zeros (all chunks equally long)
        i = torch.arange(n, device=device, dtype=torch.int64).repeat_interleave(ef)
        j = torch.arange(ef, device=device, dtype=torch.int64).repeat(n)
        shift = torch.randint(0, n // ef, (ef,), generator=g, device=device, dtype=torch.int64)
        mix = (i * 2654435761) % (n // ef)
        rows, cols = i, (j * (n // ef) + (mix + shift[j]) % (n // ef)) % n
    else:
        a, b, c, _ = abcd
        rows = torch.zeros(m, dtype=torch.int64, device=device)
        cols = torch.zeros(m, dtype=torch.int64, device=device)
        for _ in range(scale):
            u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
            rbit = u >= a + b
            cbit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
            rows = (rows << 1) | rbit
            cols = (cols << 1) | cbit
            del u, rbit, cbit
    key = torch.unique(rows * n + cols)  # sorted: row-major
    del rows, cols
    rows, cols = key // n, key % n
    del key
    vals = torch.rand(rows.numel(), generator=g, device=device, dtype=dtype) + 0.5
    csr, csc = _compress(n, rows, cols, vals, device)
    return n, csr, csc


def webgoogle_device(seed, device, dtype):
    """web-Google-SHAPED pattern matrix (the real SuiteSparse file is not available offline): n = 916428,
    ~5.1 M distinct entries, heavy-tailed out- and in-degrees, values 1.0 (pattern file -> 1.0, SimSpGEMM.cpp:92-93)."""
    import torch
    n, m = 916428, 7_000_000
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    # sources: heavy-tailed out-degree; targets: 85 % "same site" (within +-6 ids), 15 % popular hubs.
    # Calibrated on the CPU against the published figures of web-Google (nnz 5.1 M, P 60.7 M, nnz(C) 29.7 M):
    # this recipe gives nnz ~5.2 M, P ~54 M, nnz(C) ~43 M.
    rows = (n * torch.rand(m, generator=g, device=device, dtype=torch.float64).pow(2.0)).long().clamp_(max=n - 1)
    local = torch.rand(m, generator=g, device=device) < 0.85
    near = (rows + torch.randint(-6, 7, (m,), generator=g, device=device)) % n
    hubs = (n * torch.rand(m, generator=g, device=device, dtype=torch.float64).pow(2.8)).long().clamp_(max=n - 1)
    cols = torch.where(local, near, hubs)
    perm = torch.randperm(n, generator=g, device=device)      # relabel: hub ids are not the low indices
    rows, cols = perm[rows], perm[cols]
    key = torch.unique(rows * n + cols)
    rows, cols = key // n, key % n
    vals = torch.ones(rows.numel(), device=device, dtype=dtype)
    csr, csc = _compress(n, rows, cols, vals, device)
    return n, csr, csc


def cage15_device(seed, device, dtype, side=172):
    """cage15-SHAPED matrix (SuiteSparse vanHeukelum/cage15: 5 154 859 rows, 99.2 M entries, 19.2 per row, a classic SpGEMM
    benchmark whose square compresses about 2:1; the file is not available offline): a side^3 periodic lattice, every vertex
    linked to itself, its 6 face and 12 edge neighbours (each kept with probability 0.85) and 3 random vertices.
    side = 172: n = 5 088 448, ~97 M entries, ~1.9 G partial products in the square.  Every output row is short (~370 partial
    products) and about half of the products are duplicates: the regime R-MAT does not cover."""
    import torch
    n = side ** 3
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    i = torch.arange(n, device=device, dtype=torch.int64)
    x, y, z = i % side, (i // side) % side, i // (side * side)
    rows, cols = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if abs(dx) + abs(dy) + abs(dz) > 2:
                    continue
                t = ((x + dx) % side) + ((y + dy) % side) * side + ((z + dz) % side) * side * side
                if dx == dy == dz == 0:
                    rows.append(i); cols.append(t)
                else:
                    keep = torch.rand(n, generator=g, device=device) < 0.85
                    rows.append(i[keep]); cols.append(t[keep])
    for _ in range(3):
        rows.append(i); cols.append(torch.randint(0, n, (n,), generator=g, device=device, dtype=torch.int64))
    key = torch.unique(torch.cat(rows) * n + torch.cat(cols))
    del rows, cols, x, y, z
    rows, cols = key // n, key % n
    del key
    vals = torch.rand(rows.numel(), generator=g, device=device, dtype=dtype) + 0.5
    csr, csc = _compress(n, rows, cols, vals, device)
    return n, csr, csc


def mtx_device(path_a, path_b, transpose_b, device, dtype, ingest):
    """Operands from MatrixMarket files (host parse by the library's reader, osp_mtx_read = readcoo): CSC(A) and CSR(B) --
    B transposed first unless told otherwise, as the reference's main() does -- as torch tensors on the device.  Both are
    embedded in n x n with n = the largest dimension (empty rows / columns do not change the product), which is what the
    rest of this benchmark is written for.  Parse times go to `ingest`."""
    import torch
    from outerspace_amd import spgemm as S

    def read(path):
        t0 = time.perf_counter()
        nr, nc, r, c, v = S.read_mtx(path)
        dt = time.perf_counter() - t0
        ingest.setdefault("host_parse", []).append(
            {"file": os.path.basename(path), "MB": os.path.getsize(path) / 1e6, "entries": int(len(r)), "seconds": dt,
             "M_entries_per_s": len(r) / dt / 1e6, "MB_per_s": os.path.getsize(path) / 1e6 / dt,
             "threads": os.environ.get("OSP_PARSE_THREADS", f"auto (<= 32, host has {os.cpu_count()} cores)")})
        return nr, nc, r, c, v
    a = read(path_a)
    b = a if (path_b is None or path_b == path_a) else read(path_b)
    b_nr, b_nc, b_r, b_c = (b[1], b[0], b[3], b[2]) if transpose_b else (b[0], b[1], b[2], b[3])
    if a[1] != b_nr:
        raise SystemExit(f"inner dimensions differ: A is {a[0]}x{a[1]}, B{'^T' if transpose_b else ''} is {b_nr}x{b_nc}")
    n = int(max(a[0], a[1], b_nc))

    def compress(rows, cols, vals, by_col):
        rows = torch.from_numpy(rows.astype(np.int64)).to(device)
        cols = torch.from_numpy(cols.astype(np.int64)).to(device)
        vals = torch.from_numpy(vals).to(device, dtype)
        seg, inner = (cols, rows) if by_col else (rows, cols)
        key = seg * n + inner
        order = torch.argsort(key)
        key = key[order]
        if key.numel() > 1 and bool((key[1:] == key[:-1]).any()):
            raise SystemExit("duplicate coordinate in the input (the reference throws 233, SimSpGEMM.cpp:49)")
        ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        ptr[1:] = torch.cumsum(torch.bincount(seg, minlength=n), 0)
        return (ptr, inner[order].to(torch.int32).contiguous(), vals[order].contiguous())
    csc = compress(a[2], a[3], a[4], True)
    csr = compress(b_r, b_c, b[4], False)
    return n, csr, csc, (a[0], a[1], b_nc)


def ingest_report(ctx, n, csr, csc, np_dtype, device, args, ingest, with_reference):
    """The ingest beside the product (reference: TIMER("Read Matrix") SimSpGEMM.cpp:844-850 = readcoo
    :55-100, TIMER("COO2CSR") :876-880 = coo2csr :102-152).
      device  COO -> CSC(A) / CSR(B) of the FULL operands by osp_spgemm_coo (two stable radix sorts and the duplicate check
              per operand), priced against its model of ~100 B per non-zero;
      host    the library's .mtx reader on the real files, or -- synthetic workloads -- on a sample file of
              --ingest-sample entries written for the purpose, with the compiled reference's readcoo / coo2csr timed on the
              same sample (one core: it is single-threaded by construction)."""
    import tempfile
    import torch
    from outerspace_amd import spgemm as S
    nnz_a, nnz_b = int(csc[0][-1]), int(csr[0][-1])
    ar = csc[1]
    ac = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int32), (csc[0][1:] - csc[0][:-1]))
    br = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int32), (csr[0][1:] - csr[0][:-1]))
    torch.cuda.synchronize()
    # one cold call, then the median of five warm ones.  The cold call pays for the sort buffers (hipMalloc of a few GB
    # of pool memory the products before it never asked for, and the driver clearing the pages): round 3's driver line
    # reported ONE cold call (31.7 ms) where every profile of the builder, taken warm, showed 8.7-9.0 ms.
    ms_all = []
    for _ in range(6):
        # (the conversions are what is timed; the product behind them is cut down to one column of k)
        res = ctx.spgemm_coo_device(np_dtype, n, n, n, nnz_a, (ar.data_ptr(), ac.data_ptr(), csc[2].data_ptr()),
                                    nnz_b, (br.data_ptr(), csr[1].data_ptr(), csr[2].data_ptr()), k_range=(0, 1))
        ms_all.append(float(res.info["ms_ingest"]))
        res.close()
    ms_cold, ms_warm = ms_all[0], sorted(ms_all[1:])
    ms = ms_warm[len(ms_warm) // 2]
    model = 100.0 * (nnz_a + nnz_b)   # ~100 B per non-zero (keys, payloads and histograms of the passes)
    ingest["device_coo_to_compressed"] = {"ms": ms, "ms_cold_first_call": ms_cold, "ms_warm_calls": ms_warm,
                                          "timing": "median of 5 warm calls (HIP events around the conversions); the first call also pays for the "
                                                    "pool's sort buffers",
                                          "nnz_a": nnz_a, "nnz_b": nnz_b, "model_bytes": model,
                                          "GBps": model / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                                          "frac_of_peak": model / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                                          "M_nnz_per_s": (nnz_a + nnz_b) / (ms * 1e-3) / 1e6 if ms > 0 else 0.0}
    del ac, br
    sample = None
    if "host_parse" not in ingest:
        # synthetic operands: a sample of A in file form (the rows at the head of the matrix), as scipy.io.mmwrite lays it out
        k = min(nnz_a, args.ingest_sample)
        rp = csr[0].cpu().numpy()
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))[:k] + 1
        cols = csr[1][:k].cpu().numpy().astype(np.int64) + 1
        vals = csr[2][:k].cpu().numpy().astype(np.float64)
        tmp = tempfile.mkdtemp(prefix="osp_ingest_")
        sample = os.path.join(tmp, "sample.mtx")
        with open(sample, "w") as f:
            f.write(f"%%MatrixMarket matrix coordinate real general\n%\n{n} {n} {k}\n")
        try:
            import pandas as pd
            pd.DataFrame({"r": rows, "c": cols, "v": vals}).to_csv(sample, sep=" ", header=False, index=False, mode="a", float_format="%.17g")
        except ImportError:
            with open(sample, "a") as f:
                np.savetxt(f, np.stack([rows, cols, vals], 1), fmt="%d %d %.17g")
        t0 = time.perf_counter()
        got = S.read_mtx(sample)
        dt = time.perf_counter() - t0
        assert len(got[2]) == k and np.array_equal(got[2], (rows - 1).astype(np.uint32)) and np.array_equal(got[4], vals)
        mb = os.path.getsize(sample) / 1e6
        ingest["host_parse"] = [{"file": f"sample of A: its first {k} entries written as .mtx text", "MB": mb, "entries": int(k), "seconds": dt,
                                 "M_entries_per_s": k / dt / 1e6, "MB_per_s": mb / dt,
                                 "threads": os.environ.get("OSP_PARSE_THREADS", f"auto (<= 32, host has {os.cpu_count()} cores)"),
                                 "full_operand_extrapolated_s": nnz_a / (k / dt)}]
    if with_reference:
        from oracle import oracle   # baseline only
        if oracle.have_ref() and sample is not None:
            r = oracle.ref(np_dtype)
            t0 = time.perf_counter()
            nr, nc, rr, cc, vv = r.readcoo(sample)
            t1 = time.perf_counter()
            r.coo2csr(True, n, rr, cc, vv)
            r.coo2csr(False, n, rr, cc, vv)
            t2 = time.perf_counter()
            k = len(rr)
            ingest["reference_on_the_same_sample"] = {"entries": int(k), "read_matrix_s": t1 - t0, "coo2csr_x2_s": t2 - t1, "cores": 1,
                                                      "M_entries_per_s_read": k / (t1 - t0) / 1e6,
                                                      "M_entries_per_s_coo2csr": 2 * k / (t2 - t1) / 1e6}
    if sample is not None:
        os.remove(sample)
        os.rmdir(os.path.dirname(sample))
    return ingest


def webgoogle_file():
    """The real SuiteSparse web-Google file, when the box has it (it is not in the repository: no network here)."""
    for cand in (os.environ.get("OSP_WEBGOOGLE_MTX"), os.path.join(ROOT, "web-Google.mtx"), "web-Google.mtx"):
        if cand and os.path.exists(cand):
            return cand
    return None


def webgoogle_operands(seed, device, dtype):
    """configs[1]: the real file's self-product when it is present, else the synthetic shape."""
    path = webgoogle_file()
    if path:
        n, csr, csc, _ = mtx_device(path, None, False, device, dtype, {})
        return n, csr, csc
    return webgoogle_device(seed, device, dtype)


def expected_value_sum(n, csr, csc, device):
    """1^T C 1 for C = A*B without forming C: (1^T A)(B 1) = sum_k colsum_A[k] * rowsum_B[k]."""
    import torch
    colsum_a = torch.zeros(n, dtype=torch.float64, device=device).index_add_(
        0, torch.repeat_interleave(torch.arange(n, device=device), csc[0][1:] - csc[0][:-1]), csc[2].double())
    rowsum_b = torch.zeros(n, dtype=torch.float64, device=device).index_add_(
        0, torch.repeat_interleave(torch.arange(n, device=device), csr[0][1:] - csr[0][:-1]), csr[2].double())
    return float((colsum_a * rowsum_b).sum())


# ---- the CPU reference on a k-slab, and the GPU checked against it ---------------------------------------------------------
def cpu_baseline(ctx, csc, csr, n, target_partials, np_dtype, ptrs, full=False):
    """Time the reference algorithm (cscMulcsr + deduplicateCOO, SimSpGEMM.cpp:265-281,:519-535) on one host core over a
    contiguous k-slab holding about `target_partials` partial products, then run the GPU on the SAME slab (k_range) and
    compare: coordinates must be identical, values within 1e-6 (f64) / 1e-5 (f32) relative -- bit-identical when the
    baseline is the oracle port, whose summation order the GPU reproduces."""
    import torch
    from oracle import oracle  # checker / baseline only
    colptr, rowidx, avals = csc
    rowptr, colidx, bvals = csr
    w = (colptr[1:] - colptr[:-1]) * (rowptr[1:] - rowptr[:-1])
    cum = torch.cumsum(w, 0)
    total = int(cum[-1])
    k1 = int(torch.searchsorted(cum, torch.tensor([int(min(target_partials, total))], device=cum.device))[0]) + 1
    k1 = n if full else max(1, min(k1, n))   # full: the whole product ("run it in full for C2 and C3-uniform")
    a1, b1 = int(colptr[k1]), int(rowptr[k1])
    ac = colptr[:k1 + 1].cpu().numpy()
    bc = rowptr[:k1 + 1].cpu().numpy()
    ai = rowidx[:a1].cpu().numpy().view(np.uint32)
    av = avals[:a1].cpu().numpy().astype(np_dtype)
    bi = colidx[:b1].cpu().numpy().view(np.uint32)
    bv = bvals[:b1].cpu().numpy().astype(np_dtype)
    if oracle.have_ref():
        kind = "reference"
        r = oracle.ref(np_dtype).spgemm_csx(k1, ac, ai, av, bc, bi, bv)
        nnzc, P, secs = r["nnzc"], r["partials"], sum(r["secs"])
        want_rowptr = np.zeros(n + 1, np.int64)
        want_rowptr[1:] = np.cumsum(np.bincount(r["rows"], minlength=n))
        want_cols, want_vals = r["cols"], r["vals"]
    else:
        kind = "port"
        r = oracle.port().spgemm(n, k1, n, ac, ai, av, bc, bi, bv)
        nnzc, P, secs = len(r["colidx"]), r["partials"], sum(r["secs"])
        want_rowptr, want_cols, want_vals = r["rowptr"], r["colidx"], r["vals"]
    out = {"value": nnzc / secs, "unit": "nnz/s", "cores": 1, "kind": kind,
           "sample": ("the WHOLE product" if full else f"k-slab [0,{k1}) of the same matrix") + f": {P} partial products -> {nnzc} nnz in {secs:.2f} s "
                     f"({P / secs / 1e6:.2f} M partials/s); host has {os.cpu_count()} cores, the reference is single-threaded",
           "partials_per_s": P / secs, "seconds": secs}
    # ---- the GPU on the same slab ----
    note(f"reference done on k-slab [0,{k1}): {P} partial products in {secs:.1f} s; the same slab on the GPU")
    res = ctx.spgemm_csc_csr_device(np_dtype, n, n, n, ptrs, validate=False, k_range=(0, k1))
    tol = 1e-6 if np.dtype(np_dtype) == np.float64 else 1e-5
    par = {"status": "ok", "k_range": [0, k1], "partials": int(res.info["partials"]), "nnz": int(res.nnz), "against": kind,
           "value_tolerance": tol}
    problems = []
    if res.info["partials"] != P:
        problems.append(f"partial products {res.info['partials']} != {P}")
    if res.nnz != nnzc or not np.array_equal(res.rowptr, want_rowptr):
        problems.append("rowptr differs")
    elif not np.array_equal(res.colidx, want_cols):
        problems.append("colidx differs")
    else:
        got = res.vals
        err = np.abs(got - want_vals) / np.maximum(np.abs(want_vals), np.finfo(np_dtype).tiny)
        par["max_rel_err"] = float(err.max()) if len(err) else 0.0
        par["bit_identical_values"] = bool(np.array_equal(got, want_vals))
        if par["max_rel_err"] > tol or (kind == "port" and not par["bit_identical_values"]):
            problems.append(f"values differ by {par['max_rel_err']:.3e} relative")
        del err, got
    res.close()
    del want_rowptr, want_cols, want_vals, r
    if problems:
        par["status"] = "MISMATCH: " + "; ".join(problems)
    return out, par


def cpu_baseline_all_cores(csc, csr, n, np_dtype, threads, partials_each=4e7):
    """BASELINE.md section 2's optional all-core line, labelled separately: the reference is single-threaded by construction, but
    the outer product is k-separable -- `threads` host threads run the reference on `threads` disjoint k-slabs of about
    `partials_each` partial products at the same time (ctypes releases the GIL; the slabs' results are not merged with each
    other).  Aggregate output nnz/s over the wall time of the slowest."""
    import threading
    import torch
    from oracle import oracle  # baseline only
    if not oracle.have_ref():
        return None
    colptr, rowidx, avals = csc
    rowptr, colidx, bvals = csr
    cum = torch.cumsum((colptr[1:] - colptr[:-1]) * (rowptr[1:] - rowptr[:-1]), 0)
    total = int(cum[-1])
    # slabs from the middle of the k range on (the first columns are the hub columns: one of them alone is 1e8 products)
    start = int(cum[n // 3])
    marks = torch.tensor([min(start + int(partials_each) * i, total) for i in range(threads + 1)], device=cum.device)
    ks = (torch.searchsorted(cum, marks) + 1).clamp_(max=n).tolist()
    slabs = []
    for k0, k1 in zip(ks[:-1], ks[1:]):
        if k1 <= k0:
            continue
        a0, a1, b0, b1 = int(colptr[k0]), int(colptr[k1]), int(rowptr[k0]), int(rowptr[k1])
        slabs.append(((colptr[k0:k1 + 1] - a0).cpu().numpy(), rowidx[a0:a1].cpu().numpy().view(np.uint32), avals[a0:a1].cpu().numpy().astype(np_dtype),
                      (rowptr[k0:k1 + 1] - b0).cpu().numpy(), colidx[b0:b1].cpu().numpy().view(np.uint32), bvals[b0:b1].cpu().numpy().astype(np_dtype), k1 - k0))
    res = [None] * len(slabs)

    def work(i):
        ac, ai, av, bc, bi, bv, K = slabs[i]
        r = oracle.ref(np_dtype).spgemm_csx(K, ac, ai, av, bc, bi, bv, timing_only=True)
        res[i] = (int(r["nnzc"]), int(r["partials"]))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(slabs))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    secs = time.perf_counter() - t0
    if any(r is None for r in res):
        return None
    nnz, P = sum(r[0] for r in res), sum(r[1] for r in res)
    return {"value": nnz / secs, "unit": "nnz/s", "cores": len(slabs), "kind": "reference",
            "sample": f"{len(slabs)} disjoint k-slabs of the same matrix at once, one host thread each ({P} partial products -> {nnz} nnz in {secs:.2f} s; "
                      f"the reference is single-threaded: this is the k-separable outer product run side by side, the slabs' results not merged); "
                      f"host has {os.cpu_count()} cores", "partials_per_s": P / secs, "seconds": secs}


def row_slab_parity(ctx, csc, csr, n, target_partials, np_dtype, ptrs, device):
    """Parity of the WHOLE product on a row slab: the product is computed once more and kept; output rows [0, r1) -- about
    `target_partials` partial products over ALL k, the heaviest rows of an R-MAT matrix (the hub row, planned long rows, short
    rows) -- are compared with the CPU reference run on A restricted to those rows: entry counts per row and columns exactly,
    values within the tolerance (bit for bit against the oracle port).  A k-slab has almost no duplicate keys per row; this is
    the long-row merge (deduplicateCOO, SimSpGEMM.cpp:519-535) at full size."""
    import torch
    from oracle import oracle  # checker only
    from outerspace_amd.distributed import _as_tensor
    colptr, rowidx, avals = csc
    rowptr, colidx, bvals = csr
    kcol = torch.repeat_interleave(torch.arange(n, device=device), colptr[1:] - colptr[:-1])
    U = torch.zeros(n, dtype=torch.int64, device=device).index_add_(0, rowidx.long(), (rowptr[1:] - rowptr[:-1])[kcol])
    r1 = min(n, int(torch.searchsorted(torch.cumsum(U, 0), torch.tensor([int(target_partials)], device=device))[0]) + 1)
    keep = rowidx.long() < r1
    ac = np.zeros(n + 1, np.int64)
    ac[1:] = torch.cumsum(torch.zeros(n, dtype=torch.int64, device=device).index_add_(0, kcol[keep], torch.ones_like(kcol[keep])), 0).cpu().numpy()
    ai = rowidx[keep].cpu().numpy().view(np.uint32)
    av = avals[keep].cpu().numpy().astype(np_dtype)
    bc, bi, bv = rowptr.cpu().numpy(), colidx.cpu().numpy().view(np.uint32), bvals.cpu().numpy().astype(np_dtype)
    P = int(U[:r1].sum())
    del kcol, U, keep
    if oracle.have_ref():
        kind = "reference"
        r = oracle.ref(np_dtype).spgemm_csx(n, ac, ai, av, bc, bi, bv)
        want_rowptr = np.zeros(n + 1, np.int64)
        want_rowptr[1:] = np.cumsum(np.bincount(r["rows"], minlength=n))
        want_cols, want_vals, Pr = r["cols"], r["vals"], r["partials"]
    else:
        kind = "port"
        r = oracle.port().spgemm(n, n, n, ac, ai, av, bc, bi, bv)
        want_rowptr, want_cols, want_vals, Pr = r["rowptr"], r["colidx"], r["vals"], r["partials"]
    note(f"row slab [0,{r1}): reference done on {Pr} partial products; the whole product on the GPU once more")
    res = ctx.spgemm_csc_csr_device(np_dtype, n, n, n, ptrs, validate=False)
    ctx.trim()   # the pool holds the product's staging memory: torch needs room for the comparison (the result stays)
    tol = 1e-6 if np.dtype(np_dtype) == np.float64 else 1e-5
    par = {"status": "ok", "rows": [0, r1], "partials": Pr, "against": kind, "value_tolerance": tol,
           "what": "rows of the WHOLE product (all k) against the CPU reference on A restricted to them"}
    tdt, vt = (torch.float64, "<f8") if np.dtype(np_dtype) == np.float64 else (torch.float32, "<f4")
    rp, ci, va = res.device_ptrs()
    got_rowptr = _as_tensor(rp, n + 1, "<i8", device, torch.int64)[:r1 + 1]
    hi = int(got_rowptr[-1])
    problems = []
    if Pr != P:
        problems.append(f"partial products {Pr} != {P}")
    if hi != len(want_cols) or not torch.equal(got_rowptr, torch.from_numpy(want_rowptr[:r1 + 1]).to(device)):
        problems.append("rowptr differs")
    elif not torch.equal(_as_tensor(ci, hi, "<i4", device, torch.int32), torch.from_numpy(want_cols.view(np.int32)).to(device)):
        problems.append("colidx differs")
    else:
        got = _as_tensor(va, hi, vt, device, tdt)
        want = torch.from_numpy(want_vals).to(device)
        par["nnz"] = hi
        par["bit_identical_values"] = bool(torch.equal(got, want))
        par["max_rel_err"] = float(((got - want).abs() / want.abs().clamp_min(torch.finfo(tdt).tiny)).max()) if hi else 0.0
        if par["max_rel_err"] > tol or (kind == "port" and not par["bit_identical_values"]):
            problems.append(f"values differ by {par['max_rel_err']:.3e} relative")
        del got, want
    res.close()
    if problems:
        par["status"] = "MISMATCH: " + "; ".join(problems)
    return par


# ---- single-GPU measurement ------------------------------------------------------------------------------------------------
def make_step(ctx, n, csr, csc, np_dtype, tdtype, device, partial_capacity, stream_output):
    from outerspace_amd.distributed import _as_tensor
    import torch
    ptrs = [t.data_ptr() for t in (*csc, *csr)]
    vt = "<f8" if np.dtype(np_dtype) == np.float64 else "<f4"
    if stream_output:
        # C never resident: every finished row panel is checksummed on the device and dropped (the way to run
        # products whose result does not fit, e.g. Graph500 parameters at scale 22)
        def step(checksum=False):
            acc = {"sum": 0.0, "nnz": 0}

            def on_panel(p):
                acc["nnz"] += p["nnz"]
                if checksum and p["nnz"]:
                    acc["sum"] += float(_as_tensor(p["vals"], p["nnz"], vt, device, tdtype).sum(dtype=torch.float64))
            info = ctx.spgemm_csc_csr_panels(np_dtype, n, n, n, ptrs, on_panel, partial_capacity=partial_capacity)
            assert acc["nnz"] == info["nnz_c"]
            info["val_sum_global"] = acc["sum"]
            return info
    else:
        def step(checksum=False):
            res = ctx.spgemm_csc_csr_device(np_dtype, n, n, n, ptrs, validate=False, partial_capacity=partial_capacity)
            info = res.info
            if checksum:
                _, _, va = res.device_ptrs()
                info["val_sum_global"] = float(_as_tensor(va, res.nnz, vt, device, tdtype).sum(dtype=torch.float64))
            res.close()
            return info
    return step, ptrs


def library_id():
    """What identifies the kernels a number was measured on: a hash of the library's sources (and of the built .so)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = sorted(glob.glob(os.path.join(ROOT, "outerspace_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "outerspace_amd", "csrc", "*.hip")) +
                 glob.glob(os.path.join(ROOT, "outerspace_amd", "csrc", "*.cpp")))
    for path in src:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    out = {"kernel_source_sha16": h.hexdigest()[:16]}
    so = os.path.join(ROOT, "outerspace_amd", "libouterspace_spgemm.so")
    if os.path.exists(so):
        with open(so, "rb") as f:
            out["so_sha16"] = hashlib.sha256(f.read()).hexdigest()[:16]
    return out


def add_measured_peak(roof, copy_gbps):
    """Both fractions: of the data sheet's 8 TB/s and of what a plain copy reached on this box in this run."""
    roof["peak_measured"] = copy_gbps
    roof["peak_measured_what"] = ("16-bytes-per-lane copy of 2 GiB, one workgroup per 4 KB (read + written bytes / time), 10 launches on the library's stream in "
                                  "this run (osp_stream_copy_probe)")
    roof["frac_of_measured"] = roof["achieved"] / copy_gbps if copy_gbps else None
    for k in roof["kernels"].values():
        k["frac_of_measured"] = k["GBps"] / copy_gbps if copy_gbps else None
    return roof


def serial_steps(step, infos, n=2):
    """In a product of several panels the plan of panel p+1 runs on the context's second stream BESIDE the multiply of panel
    p: the two share the CUs, so neither's launch duration in the timed steps is that kernel's own.  For their
    roofline entries the product is run `n` more times with OSP_PLAN_OVERLAP=0 (every plan in line, before its own multiply);
    `value` and the merge kernel's entry stay those of the timed steps."""
    if not infos or not infos[-1].get("plans_overlapped"):
        return None
    old = os.environ.get("OSP_PLAN_OVERLAP")
    os.environ["OSP_PLAN_OVERLAP"] = "0"
    try:
        return [step() for _ in range(n)]
    finally:
        if old is None:
            os.environ.pop("OSP_PLAN_OVERLAP", None)
        else:
            os.environ["OSP_PLAN_OVERLAP"] = old


def kernel_roofline(infos, n, E, serial=None):
    """Per-kernel roofline: algorithmic bytes per launch / mean launch duration, from
    the HIP events the library records on its own stream around exactly those launches.  In the k-sharded product the
    multiply runs in the rank's local step and the merge in its final step (`final_info`): each kernel is priced on the
    call it ran in.  `serial`: infos of steps run without the plan/multiply overlap (serial_steps): the multiply's and the
    plans' durations come from those, the overlapped ones are kept beside them."""
    beside = None
    if serial:
        beside = {"multiply_kernel": float(np.mean([i["ms_multiply_kernel"] for i in infos])) / max(1, infos[-1]["multiply_launches"]),
                  "direct_plan_kernel": float(np.mean([i["ms_direct_plan_kernel"] for i in infos])) / max(1, infos[-1]["direct_plan_launches"]),
                  "hub_plan_kernel": float(np.mean([i["ms_hub_plan_kernel"] for i in infos])) / max(1, infos[-1]["hub_plan_launches"])}
        timed_infos, infos = infos, serial
    info = infos[-1]
    minfos = [i.get("final_info", i) for i in infos]   # where the merge (and the split) ran
    minfo = minfos[-1]
    tinfos = [i.get("final_info", i) for i in (timed_infos if serial else infos)]   # the merge: always the timed steps

    def mean(key, src=None):
        return float(np.mean([i[key] for i in (src or infos)]))
    Pl, nnz_al = info["partials"], info["nnz_a"]  # this rank's product
    kernels = {}
    nmul = max(1, info["multiply_launches"])
    nmer = max(1, minfo["merge_launches"])
    nrows = int(minfo.get("M", n))
    # Gathered rows: the merge kernel forms their partial products itself -- for the share g of the products it
    # does the multiply phase's work too, and is priced on both phases' algorithmic bytes for that share (a design
    # that does not spill partials is still scored against the two-phase figure); the multiply kernel on the rest.
    g = (minfo.get("gathered_partials", 0) + minfo.get("gathered_short_partials", 0)) / max(1, minfo["partials"])
    mul_model = E * (nnz_al + info["nnz_b"]) + 2 * 8 * (n + 1) + E * Pl
    mul_bytes = (1.0 - g) * mul_model / nmul
    mer_only = (E * minfo["partials"] + E * minfo["nnz_c"] + 8 * (nrows + 1)) / nmer
    mer_bytes = mer_only + g * mul_model / nmer
    for name, nbytes, ms, nl in (("multiply_kernel", mul_bytes, mean("ms_multiply_kernel"), nmul),
                                 ("merge_tiles_kernel", mer_bytes, mean("ms_merge_kernel", tinfos), nmer)):
        per = ms / nl
        kernels[name] = {"algorithmic_bytes_per_launch": nbytes, "ms_per_launch": per, "launches_per_step": nl,
                         "GBps": (nbytes / (per * 1e-3) / 1e9) if per > 0 else 0.0}
    if g > 0:
        k = kernels["merge_tiles_kernel"]
        k["gathered_share_of_partials"] = g
        k["what"] = ("multiply + merge of the gathered rows, merge of the rest: priced on the merge phase's bytes plus the gathered share of "
                     "the multiply phase's; merge phase alone: algorithmic_bytes_merge_phase_only / GBps_merge_phase_only")
        k["algorithmic_bytes_merge_phase_only"] = mer_only
        k["GBps_merge_phase_only"] = (mer_only / (k["ms_per_launch"] * 1e-3) / 1e9) if k["ms_per_launch"] > 0 else 0.0
        kernels["multiply_kernel"]["what"] = "the rows that are still written: the share 1 - g of the multiply phase's bytes"
    if minfo.get("split_launches"):
        # the one-workgroup split of long rows: two reads and one write of every record it moves
        nsp = minfo["split_launches"]
        per = mean("ms_split_kernel", minfos) / nsp
        nbytes = 3.0 * E * minfo["split_partials"] / nsp
        kernels["split_row_kernel"] = {"algorithmic_bytes_per_launch": nbytes, "ms_per_launch": per, "launches_per_step": nsp,
                                       "GBps": (nbytes / (per * 1e-3) / 1e9) if per > 0 else 0.0}
    if info.get("expand_launches"):
        # the long rows beyond the planner, staged row by row: B's entries read (gathered), the records written
        nx = info["expand_launches"]
        per = mean("ms_expand_kernel") / nx
        nbytes = 2.0 * E * info["expand_partials"] / nx
        kernels["expand_rows_kernel"] = {"algorithmic_bytes_per_launch": nbytes, "ms_per_launch": per, "launches_per_step": nx,
                                         "GBps": (nbytes / (per * 1e-3) / 1e9) if per > 0 else 0.0}
    if minfo.get("direct_plan_launches"):
        # the plan of the direct rows: B's column indices of every such row read twice (histogram, cells); what it writes
        # (one word per chunk and range) is small beside that
        npl = minfo["direct_plan_launches"]
        per = mean("ms_direct_plan_kernel", minfos) / npl
        nbytes = 2.0 * 4 * minfo["direct_partials"] / npl
        kernels["direct_plan_kernel"] = {"algorithmic_bytes_per_launch": nbytes, "ms_per_launch": per, "launches_per_step": npl,
                                         "GBps": (nbytes / (per * 1e-3) / 1e9) if per > 0 else 0.0}
    if minfo.get("hub_plan_launches"):
        # the plan of the hub rows: two walks over the runs of their chunks (two run-table entries and one column per run), one cell written per run
        nh = minfo["hub_plan_launches"]
        per = mean("ms_hub_plan_kernel", minfos) / nh
        nbytes = (2.0 * 12 + 4) * minfo["hub_cells"] / nh
        kernels["hub_plan_kernel"] = {"algorithmic_bytes_per_launch": nbytes, "ms_per_launch": per, "launches_per_step": nh,
                                      "GBps": (nbytes / (per * 1e-3) / 1e9) if per > 0 else 0.0,
                                      "what": "per panel: both walks, the scans between them and one read-back"}
    for k in kernels.values():
        k["frac_of_peak"] = k["GBps"] / HBM_PEAK_GBS
    if beside:
        for name, ms in beside.items():
            if name in kernels:
                kernels[name]["measured"] = (f"{len(serial)} steps with OSP_PLAN_OVERLAP=0 after the timed ones: in the timed steps the plan of "
                                             "panel p+1 runs beside the multiply of panel p and both take longer (ms_per_launch_beside)")
                kernels[name]["ms_per_launch_beside"] = ms
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_launch"] * kernels[k]["launches_per_step"])
    return {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"],
            "ms_per_launch": kernels[dom]["ms_per_launch"], "kernels": kernels}


def timed(step, steps, warmup, sync, barrier=None, reduce_max=None):
    for _ in range(warmup):
        step()
    sync()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    infos = [step() for _ in range(steps)]
    sync()
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    if reduce_max:
        dt = reduce_max(dt)
    return dt, infos


def check_sum(got, want, dtype, what):
    rel = abs(got - want) / max(abs(want), 1e-300)
    if rel > (1e-9 if dtype == "f64" else 1e-4):
        raise SystemExit(f"RESULT CHECK FAILED ({what}): sum(C) = {got!r}, expected {want!r} (rel {rel:.3e})")
    return {"sum_C": got, "expected_(1^T A)(B 1)": want, "rel_err": rel}


def extra_workload(ctx, name, n, csr, csc, args, np_dtype, tdtype, device, E, stream, steps=3, copy_gbps=None, cpu_full=False):
    """One secondary workload, measured the same way as the headline (fewer steps) and checked the same way.  cpu_full: the
    CPU reference runs the WHOLE product beside it and the GPU result is compared with it entry by entry."""
    import torch
    torch.cuda.synchronize()   # the operands come from torch kernels on ANOTHER stream than the library's: they must be complete
    step, ptrs = make_step(ctx, n, csr, csc, np_dtype, tdtype, device, 0, stream)
    dt, infos = timed(step, steps, 1, torch.cuda.synchronize)
    info = infos[-1]
    chk = check_sum(step(checksum=True)["val_sum_global"], expected_value_sum(n, csr, csc, device), args.dtype, name)
    ms = dt / steps * 1e3
    roof = kernel_roofline(infos, n, E, serial_steps(step, infos, 1))
    alg = 2 * E * info["partials"] + E * (2 * info["nnz_a"] + info["nnz_c"]) + 8 * (3 * n + 3)
    rec = {"ms_per_step": ms, "steps": steps, "value": info["nnz_c"] / (ms * 1e-3), "unit": "nnz/s",
           "partials_per_s": info["partials"] / (ms * 1e-3), "n": n, "nnz_a": int(info["nnz_a"]), "partials": int(info["partials"]),
           "nnz_c": int(info["nnz_c"]), "panels": int(info["panels"]), "streamed": bool(stream),
           "long_row_partials_direct": int(info["direct_partials"]), "long_row_partials_hub": int(info["hub_partials"]),
           "whole_product_GBps_algorithmic": alg / (ms * 1e-3) / 1e9, "whole_product_frac_of_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "whole_product_frac_of_measured": (alg / (ms * 1e-3) / 1e9 / copy_gbps) if copy_gbps else None,
           "phases_ms": {k: float(np.mean([i[k] for i in infos])) for k in ("ms_symbolic", "ms_multiply", "ms_merge", "ms_compact", "ms_total")},
           "kernels": {k: {"GBps": v["GBps"], "frac_of_peak": v["frac_of_peak"], "frac_of_measured": (v["GBps"] / copy_gbps) if copy_gbps else None,
                           "ms_per_launch": v["ms_per_launch"], "launches_per_step": v["launches_per_step"],
                           **({"ms_per_launch_beside": v["ms_per_launch_beside"]} if "ms_per_launch_beside" in v else {})} for k, v in roof["kernels"].items()},
           "result_check_rel_err": chk["rel_err"]}
    if cpu_full:
        note(f"{name}: the CPU reference on the whole product ({info['partials']} partial products)")
        rec["cpu_baseline"], rec["whole_product_parity"] = cpu_baseline(ctx, csc, csr, n, float("inf"), np_dtype, ptrs, full=True)
        rec["speedup_vs_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
        note(f"{name}: CPU {rec['cpu_baseline']['value'] / 1e6:.1f} M nnz/s in {rec['cpu_baseline']['seconds']:.1f} s; whole-product parity: "
             f"{rec['whole_product_parity']['status']}")
    return rec


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner and gloo a connection notice there (from C,
    not through sys.stdout): keep the real stdout for the line and point descriptor 1 at stderr for everybody else."""
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    os.write(_JSON_FD if _JSON_FD is not None else 1, line)


def library_multi_child(args, world):
    """Run `bench.py --library-multi-only WORLD` as a child process and return its JSON (or an error record)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR",
                                                             "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "OSP_BENCH_SPAWNED")}
    cmd = [sys.executable, os.path.abspath(__file__), "--library-multi-only", str(world), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--scale", str(args.scale), "--edge-factor", str(args.edge_factor), "--rmat", args.rmat, "--seed",
           str(args.seed), "--dtype", args.dtype, "--workload", args.workload]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=args.library_multi_timeout)
        lines = r.stdout.decode(errors="replace").strip().splitlines()
        if r.returncode != 0 or not lines:
            return {"error": f"child exited with status {r.returncode}"}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {args.library_multi_timeout} s"}
    except (OSError, ValueError) as e:
        return {"error": f"{type(e).__name__}: {e}"}


def library_multi_only(args):
    """Child mode: the library's own multi-GPU product (osp_spgemm_multi) over `--library-multi-only` ranks spread over the
    visible GPUs, K timed products with the slabs resident; one JSON object on stdout."""
    import torch
    from outerspace_amd import generators as gen
    from outerspace_amd import spgemm as S
    world = args.library_multi_only
    ndev = torch.cuda.device_count()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    abcd = gen.RMAT_PRESETS[args.rmat] if args.rmat in gen.RMAT_PRESETS else tuple(float(x) for x in args.rmat.split(","))
    if args.workload == "webgoogle":
        n, csr, csc = webgoogle_operands(args.seed, device, tdtype)
    else:
        n, csr, csc = rmat_device(args.scale, args.edge_factor, abcd, args.seed, device, tdtype)
    want_sum = expected_value_sum(n, csr, csc, device)
    host = [t.cpu().numpy() for t in (*csc, *csr)]
    host[1] = host[1].view(np.uint32)
    host[4] = host[4].view(np.uint32)
    del csr, csc
    torch.cuda.empty_cache()
    mg = S.MultiGpu([g % ndev for g in range(world)])
    mg.load(n, n, n, *host)
    del host
    for _ in range(args.warmup):
        mg.multiply(fetch=False)
    t0 = time.perf_counter()
    infos = [mg.multiply(fetch=False)[0] for _ in range(args.steps)]
    dt = time.perf_counter() - t0
    chk = check_sum(mg.multiply(fetch=False, checksum=True)[0]["val_sum"], want_sum, args.dtype, "library multi-GPU")
    li = infos[-1]
    ms = dt / args.steps * 1e3
    out = {"value": li["nnz_c"] / (ms * 1e-3), "ms_per_step": ms, "nnz_c": li["nnz_c"], "partials": li["partials"], "result_check": chk,
           "subpanels": li["subpanels"], "bytes_exchanged": li["bytes_exchanged"], "ms_upload_once": li["ms_upload"],
           "devices": [r["device"] for r in li["ranks"]], "gpus_visible": ndev,
           "ranks": [{k: r[k] for k in ("partials_local", "records_received", "bytes_sent", "bytes_to", "nnz_c", "ms_symbolic",
                                        "ms_multiply_kernel", "ms_merge", "ms_exchange", "ms_total", "copy_streams",
                                        "max_copies_outstanding", "max_copies_in_flight")} for r in li["ranks"]],
           "parallelism": f"k-sharded over {world} ranks inside the library (one process, one host thread per rank): partial products "
                          "copied GPU to GPU behind the multiply, one copy stream per destination (up to N-1 copies per rank in flight), "
                          "every row range merged on a stream of its own as its pieces arrive"}
    mg.close()
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.library_multi_only:
        return library_multi_only(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))   # before anything here touches the GPU
    claim_stdout()

    import torch
    from outerspace_amd import generators as gen
    from outerspace_amd import spgemm as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ..., or plain "
                         f"`python bench.py --gpus {args.gpus}`, which spawns them)")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("no GPU visible: this benchmark has no CPU path")
    if world > ndev and args.dist_backend == "nccl":
        raise SystemExit(f"{world} ranks but {ndev} GPU(s) visible: RCCL needs one GPU per rank (--dist-backend gloo rehearses on fewer)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if world == 1 and "RANK" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    abcd = "regular" if args.rmat == "regular" else (
        gen.RMAT_PRESETS[args.rmat] if args.rmat in gen.RMAT_PRESETS else tuple(float(x) for x in args.rmat.split(",")))
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    E = 4 + np.dtype(np_dtype).itemsize

    ingest = {}
    data_kind = "synthetic"
    a_mtx = args.a_mtx
    if a_mtx is None and args.workload == "webgoogle" and webgoogle_file():   # configs[1]: the real file when the box has it
        a_mtx = webgoogle_file()
        args.no_transpose_b = True   # the self-product A * A of configs[1]
    if a_mtx is not None:
        n, csr, csc, dims = mtx_device(a_mtx, args.b_mtx, not args.no_transpose_b, device, tdtype, ingest)
        bname = os.path.basename(args.b_mtx or a_mtx)
        workload_name = (f"files: {os.path.basename(a_mtx)} ({dims[0]}x{dims[1]}) * {bname}{'' if args.no_transpose_b else '^T'} "
                         f"-> {dims[0]}x{dims[2]}, CSC x CSR -> CSR")
        data_kind = "files"
    elif args.workload == "webgoogle":
        n, csr, csc = webgoogle_device(args.seed, device, tdtype)
        workload_name = "web-Google-shaped synthetic pattern matrix (916428 vertices, power-law degrees), self-product"
    elif args.workload == "cage15":
        n, csr, csc = cage15_device(args.seed, device, tdtype)
        workload_name = "cage15-shaped synthetic matrix (172^3 lattice vertices, 19-point stencil kept at 0.85 + 3 random links), self-product"
    else:
        n, csr, csc = rmat_device(args.scale, args.edge_factor, abcd, args.seed, device, tdtype)
        workload_name = (f"R-MAT scale-{args.scale} edge-factor-{args.edge_factor} (a,b,c,d)={abcd} seed {args.seed}, "
                         f"duplicates removed, self-product C=A*A, CSC x CSR -> CSR")
    nnz_a = int(csr[0][-1])
    note(f"operands on the device: n = {n}, nnz = {nnz_a}")
    want_sum = expected_value_sum(n, csr, csc, device)
    torch.cuda.synchronize()
    ctx = S.Context(dev_index)
    ctx.algorithm = args.algorithm
    cdev = device if args.dist_backend == "nccl" else "cpu"

    def reduce_max(x):
        t = torch.tensor([x], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    out = {"metric": "spgemm_output_nnz_per_s", "unit": "nnz/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": data_kind}

    if not use_dist:
        # ================================================= one GPU =================================================
        # what a plain copy reaches on this box, in this run (the pool's 4 GiB go back before the product sizes its buffers)
        copy_gbps = ctx.stream_copy_gbps()
        ctx.trim()
        note(f"stream copy: {copy_gbps:.0f} GB/s (read + written)")
        step, ptrs = make_step(ctx, n, csr, csc, np_dtype, tdtype, device, args.partial_capacity, args.stream_output)
        dt, infos = timed(step, args.steps, args.warmup, torch.cuda.synchronize)
        info = infos[-1]
        note(f"{args.steps} timed steps: {dt / args.steps * 1e3:.1f} ms per step")
        chk = check_sum(step(checksum=True)["val_sum_global"], want_sum, args.dtype, "single GPU")
        note("whole-result check passed")
        nnz_c, P = info["nnz_c"], info["partials"]
        ms_step = dt / args.steps * 1e3
        serial = serial_steps(step, infos)
        roof = add_measured_peak(kernel_roofline(infos, n, E, serial), copy_gbps)
        alg_total = 2 * E * P + E * (2 * nnz_a + nnz_c) + 8 * (3 * n + 3)   # whole product
        out.update({
            "value": nnz_c / (ms_step * 1e-3), "ms_per_step": ms_step,
            "config": {"workload": workload_name, "n": n, "nnz_a": nnz_a, "partials": P, "nnz_c": nnz_c, "algorithm": args.algorithm,
                       "parallelism": "single GPU, output streamed panel by panel (never resident)" if args.stream_output else "single GPU"},
            "gflops": 2 * P / (ms_step * 1e-3) / 1e9, "partials_per_s": P / (ms_step * 1e-3),
            "whole_product": {"algorithmic_bytes": alg_total, "GBps": alg_total / (ms_step * 1e-3) / 1e9,
                              "frac_of_peak": alg_total / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "frac_of_measured": alg_total / (ms_step * 1e-3) / 1e9 / copy_gbps},
            "library_id": library_id(),
            "phases_ms": {k: float(np.mean([i[k] for i in infos])) for k in ("ms_symbolic", "ms_multiply", "ms_merge", "ms_compact", "ms_total")},
            "panels": info["panels"], "plans_beside_the_previous_multiply": info["plans_overlapped"],
            "ms_per_step_plans_in_line": float(np.mean([i["ms_total"] for i in serial])) if serial else None,
            "long_rows": info["heavy_rows"], "long_row_partials": info["heavy_partials"],
            # long rows the multiply wrote straight into their column ranges (no split pass), and the ones split afterwards
            "long_rows_direct": info["direct_rows"], "long_row_partials_direct": info["direct_partials"],
            # of those, the rows that were never written: the merge kernel formed their partial products from the plan's run
            # descriptors; and the short rows formed the same way
            "long_rows_gathered": info["gathered_rows"], "long_row_partials_gathered": info["gathered_partials"],
            "run_descriptors": info["gathered_runs"], "short_row_partials_gathered": info["gathered_short_partials"],
            # rows beyond the one-workgroup planner that the multiply wrote into uniform column blocks (no stretch split)
            "long_rows_hub": info["hub_rows"], "long_row_partials_hub": info["hub_partials"], "hub_cells": info["hub_cells"],
            "long_row_partials_split_by_one_workgroup": info["split_partials"],
            "segments_global_sorted": info["sorted_segments"], "segment_partials_global_sorted": info["sorted_partials"],
            "roofline": roof,
            # per-step device times (ms): how much the numbers above move from one product to the next
            "steps_ms": {"total": [round(i["ms_total"], 2) for i in infos],
                         "multiply_kernel": [round(i["ms_multiply_kernel"], 2) for i in infos],
                         "merge_kernel": [round(i["ms_merge_kernel"], 2) for i in infos],
                         "split_row_kernel": [round(i["ms_split_kernel"], 2) for i in infos],
                         "direct_plan_kernel": [round(i["ms_direct_plan_kernel"], 2) for i in infos]},
            "result_check": chk,
        })
        status = 0
        if args.cpu_baseline:
            out["cpu_baseline"], out["slab_parity"] = cpu_baseline(ctx, csc, csr, n, args.cpu_partials, np_dtype, ptrs)
            note(f"CPU baseline {out['cpu_baseline']['value'] / 1e6:.1f} M nnz/s; slab parity: {out['slab_parity']['status']}")
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            if out["slab_parity"]["status"] != "ok":
                status = 3
            if not args.stream_output and args.workload == "rmat" and not args.a_mtx:
                try:
                    allc = cpu_baseline_all_cores(csc, csr, n, np_dtype, min(16, os.cpu_count() or 1))
                except Exception as e:   # (an optional line: never at the price of the run)
                    allc = {"error": repr(e)}
                if allc:
                    out["cpu_baseline_all_cores"] = allc
                    if "value" in allc:
                        note(f"CPU reference on {allc['cores']} threads side by side: {allc['value'] / 1e6:.1f} M nnz/s")
                out["row_slab_parity"] = row_slab_parity(ctx, csc, csr, n, 5e7, np_dtype, ptrs, device)
                note(f"row slab parity: {out['row_slab_parity']['status']}")
                if out["row_slab_parity"]["status"] != "ok":
                    status = 3
                ctx.trim()
        if args.ingest and not args.stream_output:
            note("ingest: device COO -> CSC/CSR of the full operands, host parse")
            out["ingest"] = ingest_report(ctx, n, csr, csc, np_dtype, device, args, ingest, bool(args.cpu_baseline))
            note(f"ingest: device {out['ingest']['device_coo_to_compressed']['ms']:.1f} ms, host parse "
                 f"{out['ingest']['host_parse'][0]['M_entries_per_s']:.1f} M entries/s")
        # the reference's own closed-form prediction for this input (SimOuterSPACE.cpp:176-238), beside the
        # measurement: simulated cycles of a 256-PE OuterSPACE at 85 B/cycle of DRAM, and the DRAM bytes it prices
        from outerspace_amd import cost_model
        torch.cuda.synchronize()
        ctx.trim()   # the library's pool may hold every free byte of the device: torch needs a few hundred MB for this
        pred = cost_model.analytical(csc[0], csc[1], csr[0], value_size=np.dtype(np_dtype).itemsize)
        note("cost model evaluated")
        pred["note"] = ("OuterSPACE analytical model restated from the reference (not a measurement): cycles of the "
                        "simulated accelerator, 64-B-aligned DRAM bytes per task")
        out["cost_model"] = pred
        # HBM bytes per launch of the dominant kernel from the PMC counters: they cannot be collected from inside
        # this process (rocprofv3 --pmc wraps it, separate passes), so the figure recorded for exactly this
        # workload under profiles/ is attached when there is one; otherwise null
        t = recorded_traffic(workload_name, args.dtype, roof["kernel"])
        if t:
            roof["traffic"], roof["traffic_source"], roof["traffic_library_id"] = t
            # a recorded figure, not a property of this run: stale when the kernels have changed since the PMC passes
            # (the same binary, or a build of the same sources: either hash matching means the counters belong to these kernels)
            rec_id, cur_id = roof["traffic_library_id"] or {}, out["library_id"]
            roof["traffic_stale"] = not any(rec_id.get(k) is not None and rec_id.get(k) == cur_id.get(k) for k in ("kernel_source_sha16", "so_sha16"))
        default_workload = (args.workload == "rmat" and args.rmat == "mild" and args.scale == 22 and not args.stream_output
                            and args.partial_capacity == 0 and a_mtx is None)
        if args.extras and default_workload and status == 0:
            # driver-observed numbers for the workloads the headline does not cover (3 steps each, same checks)
            del csr, csc, step, ptrs
            ctx.trim()
            torch.cuda.empty_cache()
            extras = {}
            for name, make, stream in (
                    ("rmat22_g500_streamed", lambda: rmat_device(22, 16, gen.RMAT_PRESETS["g500"], args.seed, device, tdtype), True),
                    ("rmat20_g500_streamed", lambda: rmat_device(20, 16, gen.RMAT_PRESETS["g500"], args.seed, device, tdtype), True),
                    ("rmat22_uniform", lambda: rmat_device(22, 16, gen.RMAT_PRESETS["uniform"], args.seed, device, tdtype), False),
                    ("cage15_shape", lambda: cage15_device(args.seed, device, tdtype), False),
                    ("webgoogle_shape", lambda: webgoogle_operands(args.seed, device, tdtype), False)):
                note(f"extra workload {name}")
                n2, csr2, csc2 = make()
                extras[name] = extra_workload(ctx, name, n2, csr2, csc2, args, np_dtype, tdtype, device, E, stream,
                                              steps={"webgoogle_shape": 5, "rmat22_g500_streamed": 2}.get(name, 3), copy_gbps=copy_gbps,
                                              cpu_full=bool(args.cpu_baseline) and name in ("webgoogle_shape", "rmat22_uniform"))
                if extras[name].get("whole_product_parity", {}).get("status", "ok") != "ok":
                    status = 3
                del csr2, csc2
                ctx.trim()
                torch.cuda.empty_cache()
            out["extra_workloads"] = extras
        emit(out)
        ctx.close()
        sys.exit(status)

    # ===================================================== N ranks =====================================================
    from outerspace_amd import distributed as D
    modes = ["k", "rows"] if args.shard == "both" else [args.shard]
    results = {}
    sync = torch.cuda.synchronize
    for mode in modes:
        if mode == "k":
            k_bounds = D.plan_k_shards(csc[0], csr[0], world)
            # this rank's operands: ONLY its columns of A and rows of B
            slab = D.slice_k_slab(csc, csr, k_bounds[rank], k_bounds[rank + 1])
            torch.cuda.synchronize()   # (sliced by torch kernels; the library works on a stream of its own)
            # the form of the exchange is a property of the slab: agreed on once, outside the timed steps
            k_exchange = D.agree_k_exchange(ctx, np_dtype, slab, dist, args.k_exchange, args.dist_backend == "gloo")

            def step(checksum=False):
                return D.spgemm_k_sharded(ctx, np_dtype, n, n, slab, dist, rank, world, partial_capacity=args.partial_capacity,
                                          stage_through_host=args.dist_backend == "gloo", checksum=checksum, exchange=k_exchange,
                                          agreed=True)
        else:
            ptrs = [t.data_ptr() for t in (*csc, *csr)]

            def step(checksum=False):
                return D.spgemm_row_sharded(ctx, np_dtype, n, n, n, ptrs, dist, rank, world, device, partial_capacity=args.partial_capacity,
                                            host_collectives=args.dist_backend == "gloo", checksum=checksum)
        dt, infos = timed(step, args.steps, args.warmup, sync, dist.barrier, reduce_max)
        note(f"{mode}-sharded over {world} ranks: {dt / args.steps * 1e3:.1f} ms per step")
        chk = check_sum(step(checksum=True)["val_sum_global"], want_sum, args.dtype, f"{mode}-sharded")
        info = infos[-1]
        ms_step = dt / args.steps * 1e3
        r = {"value": info["nnz_c_global"] / (ms_step * 1e-3), "ms_per_step": ms_step, "nnz_c": info["nnz_c_global"],
             "partials": info["partials_global"], "result_check": chk,
             "rank0_phases_ms": {k: float(np.mean([i[k] for i in infos])) for k in
                                 ("ms_symbolic", "ms_multiply", "ms_merge", "ms_total", "ms_local", "ms_exchange", "ms_final_merge") if k in info},
             "rank0_roofline": kernel_roofline(infos, n, E)}
        if mode == "k":
            sent = torch.tensor([info["bytes_sent"]], device=cdev, dtype=torch.int64)
            allsent = [torch.zeros_like(sent) for _ in range(world)]
            dist.all_gather(allsent, sent)
            r.update(k_bounds=k_bounds, bytes_sent_per_rank=[int(x[0]) for x in allsent], exchange=info.get("exchange"),
                     row_bounds=info["row_bounds"], local_nnz_rank0=info["nnz_c"], final_merge_partials_rank0=info["final_merge_partials"],
                     parallelism=f"k-sharded over {world} GPUs: each rank holds only its columns of A / rows of B, "
                                 + ("multiplies, and sends its partial products unmerged" if info.get("exchange") == "raw"
                                    else "forms its partial CSR") +
                                 ": one all-to-all-v over RCCL, one merge per output-row range (result row-sharded)")
            del slab
        else:
            r["parallelism"] = f"output rows sharded over {world} GPUs (operands replicated, result row-sharded, no exchange)"
        results[mode] = r
        ctx.trim()
        torch.cuda.empty_cache()
    if args.library_multi and "k" in modes:
        # The same decomposition INSIDE the library (include/outerspace_spgemm.h, osp_multi_*): ONE process drives all N GPUs --
        # slabs resident on their GPUs, pipelined exchange, merge overlapped with it.  It runs in a child process of rank 0
        # with a time limit (this path has never run on more than one physical GPU at the builder's: whatever it does there
        # must not cost the line above its numbers); the other ranks have released their pools and wait at the barrier.
        dist.barrier()
        if rank == 0:
            results["k_library"] = library_multi_child(args, world)
            note(f"library multi-GPU product over {world} ranks: {results['k_library'].get('ms_per_step', results['k_library'])}")
        dist.barrier()
    # what the ranks of this job saw of each other: the collective backend's world and every rank's device
    mine = torch.tensor([dev_index, ndev], device=cdev, dtype=torch.int64)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    fabric = {"backend": args.dist_backend + (" (RCCL)" if args.dist_backend == "nccl" else ""), "world_size": dist.get_world_size(),
              "rank_devices": [int(x[0]) for x in seen], "gpus_visible_per_rank": [int(x[1]) for x in seen]}
    torch_head = results["k"] if "k" in results else results[modes[0]]
    lib = results.get("k_library")
    # The headline at N > 1 is the library's own k-sharded product (one process, pipelined exchange over one stream per link)
    # when it ran and its whole-result check passed; otherwise the same decomposition over torch.distributed.  Both stay
    # under "decompositions".
    lib_ok = isinstance(lib, dict) and "error" not in lib and "result_check" in lib and lib.get("nnz_c") == torch_head["nnz_c"]
    if rank == 0:
        head = torch_head
        out.update({
            "value": head["value"], "ms_per_step": head["ms_per_step"],
            "config": {"workload": workload_name, "n": n, "nnz_a": nnz_a, "partials": head["partials"], "nnz_c": head["nnz_c"],
                       "algorithm": args.algorithm, "parallelism": head["parallelism"], "backend": args.dist_backend},
            "gflops": 2 * head["partials"] / (head["ms_per_step"] * 1e-3) / 1e9,
            "partials_per_s": head["partials"] / (head["ms_per_step"] * 1e-3),
            "phases_ms": head["rank0_phases_ms"], "roofline": head["rank0_roofline"], "result_check": head["result_check"],
            # which decomposition `value` is (the other ones, when measured, are under decompositions)
            "shard": "k" if head is results.get("k") else modes[0],
            "ms_exchange": head["rank0_phases_ms"].get("ms_exchange"), "ms_final_merge": head["rank0_phases_ms"].get("ms_final_merge"),
            "bytes_sent_per_rank": head.get("bytes_sent_per_rank"),
            "fabric": fabric,
            "decompositions": results,
            "headline_rule": ("value = the k-split north_star names: the library's own product (shard k_library) when its child process ran and "
                              "its whole-result check passed, else the same decomposition over torch.distributed (shard k); the row-sharded product "
                              "(no exchange) is under decompositions.rows.  DESIGN.md section 5's model of the k-split: the exchange ships every "
                              "partial product once -- about 560 / 140 / 40-45 ms at N = 2 / 4 / 8 for this workload against 176 ms on one GPU, i.e. x0.3, x1.3, "
                              "x4; the row-sharded product (every rank runs the one-GPU pipeline on its rows) scales from N = 2"),
        })
        if lib_ok:
            out.update({
                "value": lib["value"], "ms_per_step": lib["ms_per_step"], "shard": "k_library",
                "gflops": 2 * lib["partials"] / (lib["ms_per_step"] * 1e-3) / 1e9, "partials_per_s": lib["partials"] / (lib["ms_per_step"] * 1e-3),
                "result_check": lib["result_check"], "ms_exchange": max(r.get("ms_exchange", 0.0) for r in lib["ranks"]),
                "ms_final_merge": max(r["ms_merge"] for r in lib["ranks"]), "bytes_sent_per_rank": [r["bytes_sent"] for r in lib["ranks"]],
                "phases_ms": {k: max(r[k] for r in lib["ranks"]) for k in ("ms_symbolic", "ms_multiply_kernel", "ms_merge", "ms_total")},
                "roofline_note": "roofline: rank 0's kernels in the torch.distributed run of the same decomposition (the library's "
                                 "multi-GPU product reports per-rank phase times, not per-kernel events)",
            })
            out["config"]["parallelism"] = lib["parallelism"]
            out["config"]["backend"] = "library (hipMemcpyPeerAsync over xGMI, one stream per destination); " + args.dist_backend + " for the comparison run"
        # the fastest decomposition that ran and passed its whole-result check, beside the headline (they differ below eight GPUs)
        checked = {k: v for k, v in results.items() if isinstance(v, dict) and "error" not in v and "result_check" in v and v.get("ms_per_step")}
        if checked:
            best = min(checked, key=lambda k: checked[k]["ms_per_step"])
            out["fastest_decomposition"] = {"shard": best, "value": checked[best]["value"], "ms_per_step": checked[best]["ms_per_step"]}
        emit(out)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def recorded_traffic(workload, dtype, kernel):
    """(bytes per launch, source) from the newest profiles/*_pmc_hbm_*.json recorded for this very workload."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_hbm_*.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        k = rec.get("kernels", {}).get(kernel)
        if rec.get("workload") == workload and rec.get("dtype") == dtype and k:
            return k["traffic"], (f"{rec.get('source')}: FETCH_SIZE x2 (gfx950 correction for wide reads) + WRITE_SIZE, "
                                  f"rocprofv3 --pmc, separate passes"), rec.get("library_id")
    return None


if __name__ == "__main__":
    main()
