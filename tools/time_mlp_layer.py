#!/usr/bin/env python3
"""Times BASELINE configs[4]'s layer product (act 1024 x 784 times pruned W^T 784 x H, f32) on the device, operands resident."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from outerspace_amd import generators as gen  # noqa: E402
from outerspace_amd import spgemm as S  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "mlp_full_expected.npz"))
dev = torch.device("cuda", 0)
with S.Context(0) as ctx:
    for H in (100, 1000):
        act, W, Wp, a, b = gen.mlp_layer_operands(H, g[f"thr_{H}"])
        csc = S.coo_to_csc(784, *a)
        csr = S.coo_to_csr(784, *b)
        t = [torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) for x in (*csc, *csr)]
        ptrs = [x.data_ptr() for x in t]
        ms = []
        for it in range(8):
            r = ctx.spgemm_csc_csr_device(np.float32, 1024, 784, H, ptrs)
            i = r.info
            ms.append(i["ms_total"])
            last = i
            r.close()
        print(f"H={H}: P={last['partials']} nnzC={last['nnz_c']}  ms per product {np.median(ms[2:]):.3f}  (multiply k {last['ms_multiply_kernel']:.3f}, "
              f"merge k {last['ms_merge_kernel']:.3f}, plan k {last['ms_direct_plan_kernel']:.3f}, heavy rows {last['heavy_rows']})", flush=True)
