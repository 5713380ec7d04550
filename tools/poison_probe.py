#!/usr/bin/env python3
"""Debugging aid: one product under OSP_POISON=1; which entries of the result were never written (still 0xFF bytes)?
usage: OSP_POISON=1 python tools/poison_probe.py [scale] [preset] [seed]"""
import os, sys, importlib.util
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outerspace_amd import spgemm as S, generators as gen
from outerspace_amd.distributed import _as_tensor
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
preset = sys.argv[2] if len(sys.argv) > 2 else "mild"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
dev = torch.device("cuda", 0)
n, csr, csc = bench.rmat_device(scale, 16, gen.RMAT_PRESETS[preset], seed, dev, torch.float64)
torch.cuda.synchronize()
ptrs = [t.data_ptr() for t in (*csc, *csr)]
with S.Context(0) as ctx:
    for rep in range(3):
        r = ctx.spgemm_csc_csr_device(np.float64, n, n, n, ptrs)
        rp, ci, va = r.device_ptrs()
        rowptr = _as_tensor(rp, n + 1, "<i8", dev, torch.int64).clone()
        col = _as_tensor(ci, r.nnz, "<i4", dev, torch.int32).clone()
        val = _as_tensor(va, r.nnz, "<f8", dev, torch.float64).view(torch.int64).clone()
        bad = (val == -1).nonzero().flatten()
        badc = (col == -1).nonzero().flatten()
        print(f"rep {rep}: nnz {r.nnz}, rowptr[-1] {int(rowptr[-1])}, unwritten values {bad.numel()}, unwritten columns {badc.numel()}", {k: r.info[k] for k in ("panels", "heavy_rows", "direct_rows", "gathered_rows", "dense_segments", "sorted_segments")})
        if bad.numel():
            rows = torch.searchsorted(rowptr, bad, right=True) - 1
            print("  first/last unwritten positions", int(bad[0]), int(bad[-1]), "rows", rows[:5].tolist(), "...", rows[-5:].tolist(), "of", n)
            rr = int(rows[0]); print("  row", rr, "span", int(rowptr[rr]), int(rowptr[rr + 1]))
        r.close()
