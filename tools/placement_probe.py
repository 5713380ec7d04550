#!/usr/bin/env python3
"""Does the multiply kernel's time depend on where the pool's buffers land?  One process, the same product eight times;
the context's pool (and torch's) is released after every second product, so the staging buffers are allocated anew."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from outerspace_amd import generators as gen  # noqa: E402
from outerspace_amd import spgemm as S  # noqa: E402

dev = torch.device("cuda", 0)
n, csr, csc = bench.rmat_device(22, 16, gen.RMAT_PRESETS["mild"], 1, dev, torch.float64)
with S.Context(0) as ctx:
    ptrs = [t.data_ptr() for t in (*csc, *csr)]
    for it in range(8):
        r = ctx.spgemm_csc_csr_device(np.float64, n, n, n, ptrs)
        i = r.info
        print(f"{it}: total {i['ms_total']:.1f} ms, multiply kernels {i['ms_multiply_kernel']:.2f}, merge kernels {i['ms_merge_kernel']:.2f}, "
              f"plan {i['ms_direct_plan_kernel']:.2f}", flush=True)
        r.close()
        del r
        if it % 2 == 1:
            ctx.trim()
            torch.cuda.empty_cache()
            print("   (pool released)", flush=True)
