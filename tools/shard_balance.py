#!/usr/bin/env python3
"""How even are the row shards?  Runs every shard of a G-way row-sharded product one after the other on ONE GPU and
prints each shard's time -- the slowest one is what a G-GPU run waits for (no exchange in this mode).
usage: python tools/shard_balance.py [G] [rmat preset] [scale]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from outerspace_amd import generators as gen  # noqa: E402
from outerspace_amd import spgemm as S  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
preset = sys.argv[2] if len(sys.argv) > 2 else "mild"
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 22
dev = torch.device("cuda", 0)
n, csr, csc = bench.rmat_device(scale, 16, gen.RMAT_PRESETS[preset], 1, dev, torch.float64)
torch.cuda.synchronize()
ctx = S.Context(0)
ptrs = [t.data_ptr() for t in (*csc, *csr)]


def run(shard):
    res = ctx.spgemm_csc_csr_device(np.float64, n, n, n, ptrs, validate=False, row_shard=shard)
    info = res.info
    res.close()
    return info


run(None)
t0 = time.perf_counter(); whole = run(None); torch.cuda.synchronize(); t_whole = time.perf_counter() - t0
print(f"whole product: {t_whole * 1e3:.1f} ms, partials {whole['partials']:.4g}")
times = []
for i in range(G):
    run((i, G))
    t0 = time.perf_counter(); info = run((i, G)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    times.append(dt)
    print(f"shard {i}/{G}: rows [{info['row_begin']}, {info['row_end']}) partials {info['partials']:.4g} nnz {info['nnz_c']:.4g} "
          f"long-row partials {info['heavy_partials']:.3g}  {dt * 1e3:.1f} ms  (symbolic {info['ms_symbolic']:.1f}, multiply "
          f"{info['ms_multiply']:.1f}, merge {info['ms_merge']:.1f})")
print(f"slowest {max(times) * 1e3:.1f} ms, mean {np.mean(times) * 1e3:.1f} ms -> speed-up at {G} GPUs = {t_whole / max(times):.2f} "
      f"(perfectly even shards would give {t_whole / np.mean(times):.2f})")
ctx.close()
