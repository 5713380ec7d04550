#!/usr/bin/env python3
"""Does the multiply kernel's duration depend on WHERE the product's buffers lie?  One process, the headline product; between
trials the context's pool goes back to the driver (trim) so the next product allocates its buffers anew -- optionally with a
block of memory held in between so that they land elsewhere.  Prints per trial the in-line kernel durations of two products.
usage: tools/multiply_spread.py [trials] [scale]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from outerspace_amd import generators as gen  # noqa: E402
from outerspace_amd import spgemm as S  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 22
os.environ["OSP_PLAN_OVERLAP"] = "0"
dev = torch.device("cuda:0")
n, csr, csc = bench.rmat_device(scale, 16, gen.RMAT_PRESETS["mild"], 1, dev, torch.float64)
torch.cuda.synchronize()
ctx = S.Context(0)
step, _ = bench.make_step(ctx, n, csr, csc, np.float64, torch.float64, dev, 0, False)
hold = None
for t in range(trials):
    ctx.trim()
    torch.cuda.empty_cache()
    if t % 2 == 1:   # odd trials: 3 GiB held while the buffers are allocated, so they cannot land where they were
        hold = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
    infos = [step() for _ in range(3)]
    print(f"trial {t} hold={'yes' if hold is not None else 'no'}:", " | ".join(
        f"total {i['ms_total']:.1f} multiply {i['ms_multiply_kernel'] / 3:.2f} merge {i['ms_merge_kernel'] / 3:.2f} plan {i['ms_direct_plan_kernel'] / 3:.2f}" for i in infos), flush=True)
    hold = None
