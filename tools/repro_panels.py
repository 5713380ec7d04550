"""Streamed / multi-panel product against the one-panel product at a given staging capacity: where do they differ?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from outerspace_amd import generators as gen, spgemm
from outerspace_amd.distributed import _as_tensor

scale, preset, cap = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
dev = torch.device("cuda", 0)
n, csr, csc = bench.rmat_device(scale, 16, gen.RMAT_PRESETS[preset], 5, dev, torch.float64)
torch.cuda.synchronize()
ptrs = [t.data_ptr() for t in (*csc, *csr)]
ctx = spgemm.Context()
ref = ctx.spgemm_csc_csr_device(np.float64, n, n, n, ptrs)
nnz = ref.nnz
rp, ci, va = ref.device_ptrs()
rowptr, colidx, vals = (_as_tensor(rp, n + 1, "<i8", dev, torch.int64), _as_tensor(ci, nnz, "<i4", dev, torch.int32),
                        _as_tensor(va, nnz, "<f8", dev, torch.float64))
U = torch.zeros(n, dtype=torch.int64, device=dev)
lens = csr[0][1:] - csr[0][:-1]
kcol = torch.repeat_interleave(torch.arange(n, device=dev), csc[0][1:] - csc[0][:-1])
U.index_add_(0, csc[1].long(), lens[kcol])
for rep in range(3):
    def on_panel(p):
        lo = int(rowptr[p["row_begin"]])
        nr = p["row_end"] - p["row_begin"]
        prp = _as_tensor(p["rowptr"], nr + 1, "<i8", dev, torch.int64)
        refp = rowptr[p["row_begin"]:p["row_end"] + 1] - lo
        bad = torch.nonzero((prp[1:] - prp[:-1]) != (refp[1:] - refp[:-1])).flatten()
        print(f"rep {rep} panel {p['index']}/{p['count']} rows [{p['row_begin']},{p['row_end']}) nnz {p['nnz']} want {int(refp[-1])}: {len(bad)} rows differ")
        for r in bad[:8].tolist():
            row = p["row_begin"] + r
            print(f"   row {row}: got {int(prp[r + 1] - prp[r])} want {int(refp[r + 1] - refp[r])} partial products {int(U[row])}")
            g = _as_tensor(p["colidx"] + 4 * int(prp[r]), int(prp[r + 1] - prp[r]), "<i4", dev, torch.int32).cpu().numpy()
            wv = colidx[int(rowptr[row]):int(rowptr[row + 1])].cpu().numpy()
            miss = np.setdiff1d(wv, g)
            extra = np.setdiff1d(g, wv)
            print(f"      missing columns {miss[:10]} extra {extra[:10]}  (got sorted: {bool((np.diff(g) > 0).all())})")
    info = ctx.spgemm_csc_csr_panels(np.float64, n, n, n, ptrs, on_panel, partial_capacity=cap)
    print("info", {k: info[k] for k in ("panels", "nnz_c", "heavy_rows", "dense_segments", "sorted_segments")})
