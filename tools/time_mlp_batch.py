#!/usr/bin/env python3
"""configs[4]'s layer product at a LARGE batch: act (batch x 784, ~16 % dense) times pruned W^T (784 x H), f32, on the device.
usage: tools/time_mlp_batch.py [batch] [H] [keep]   (keep = fraction of W's entries that survive pruning)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from outerspace_amd import spgemm as S  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
keep = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
act = torch.relu(torch.randn(batch, 784, generator=g, device=dev) - 1.0)
W = torch.randn(H, 784, generator=g, device=dev) * 0.05
W = W * (torch.rand(H, 784, generator=g, device=dev) < keep)


def csc_of(m):   # dense (rows x cols) -> CSC arrays
    t = m.t().contiguous()            # (cols x rows): CSR of the transpose = CSC of m
    nz = t != 0
    ptr = torch.zeros(t.shape[0] + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(nz.sum(1), 0)
    idx = nz.nonzero()[:, 1].to(torch.int32).contiguous()
    return ptr, idx, t[nz].contiguous()


def csr_of(m):
    nz = m != 0
    ptr = torch.zeros(m.shape[0] + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(nz.sum(1), 0)
    idx = nz.nonzero()[:, 1].to(torch.int32).contiguous()
    return ptr, idx, m[nz].contiguous()


a = csc_of(act)            # A = act: batch x 784
b = csr_of(W.t().contiguous())   # B = W^T: 784 x H
torch.cuda.synchronize()
with S.Context(0) as ctx:
    ptrs = [x.data_ptr() for x in (*a, *b)]
    ms = []
    for it in range(6):
        r = ctx.spgemm_csc_csr_device(np.float32, batch, 784, H, ptrs)
        i = r.info
        ms.append(i["ms_total"])
        if it == 5:
            ref = (act.double() @ W.double().t())
            rp, ci, va = r.device_ptrs()
            from outerspace_amd.distributed import _as_tensor
            vals = _as_tensor(va, r.nnz, "<f4", dev, torch.float32)
            print(f"batch={batch} H={H} keep={keep}: nnzA={i['nnz_a']} nnzB={i['nnz_b']} P={i['partials']} nnzC={i['nnz_c']} "
                  f"ms {np.median(ms[2:]):.2f} ({np.median(ms[2:]) * 1e9 / i['partials']:.1f} ps/product)  multiply k {i['ms_multiply_kernel']:.2f} merge k {i['ms_merge_kernel']:.2f} "
                  f"plan k {i['ms_direct_plan_kernel']:.2f} heavy rows {i['heavy_rows']} direct {i['direct_rows']}  sum check {float(vals.double().sum()):.6e} vs {float(ref.sum()):.6e}", flush=True)
        r.close()
