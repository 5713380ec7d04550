"""Debugging aid: two different workloads through ONE library context, with OSP_SYNC=1 naming the phase that fails."""
import importlib.util
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("osp_bench", os.path.join(root, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
from outerspace_amd import generators as gen  # noqa: E402
from outerspace_amd import spgemm as S  # noqa: E402

dev = torch.device("cuda", 0)
ctx = S.Context(0)
seq = sys.argv[1:] or ["uniform22", "web"]
for name in seq:
    if name == "web":
        n, csr, csc = B.webgoogle_device(1, dev, torch.float64)
    elif name.startswith("uniform"):
        n, csr, csc = B.rmat_device(int(name[7:]), 16, gen.RMAT_PRESETS["uniform"], 1, dev, torch.float64)
    elif name.startswith("g500"):
        n, csr, csc = B.rmat_device(int(name[4:]), 16, gen.RMAT_PRESETS["g500"], 1, dev, torch.float64)
    else:
        n, csr, csc = B.rmat_device(int(name[4:]), 16, gen.RMAT_PRESETS["mild"], 1, dev, torch.float64)
    torch.cuda.synchronize()   # the library's stream does not wait for torch's
    print(f"== {name}: n = {n}", file=sys.stderr, flush=True)
    for rep in range(3):
        res = ctx.spgemm_csc_csr_device(np.float64, n, n, n, [t.data_ptr() for t in (*csc, *csr)], validate=False)
        print(f"== {name} rep {rep}: nnz {res.nnz}, {res.info['ms_total']:.2f} ms", file=sys.stderr, flush=True)
        res.close()
    del csr, csc
    ctx.trim()
    torch.cuda.empty_cache()
print("sequence ok")
