import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from outerspace_amd import generators as gen, spgemm as S, distributed as D
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
dev = torch.device("cuda", 0)
n, csr, csc = bench.rmat_device(scale, 16, gen.RMAT_PRESETS["mild"], 1, dev, torch.float64)
torch.cuda.synchronize()
ctx = S.Context(0)
res = ctx.spgemm_csc_csr_device(np.float64, n, n, n, [t.data_ptr() for t in (*csc, *csr)], validate=False)
print("product nnz", res.nnz, res.info["heavy_rows"], res.info["sorted_segments"])
rp, ci, va = res.device_ptrs()
m = ctx.merge_csr_parts_device(np.float64, n, n, [(rp, ci, va)])
print("single-part merge nnz", m.nnz, {k: m.info[k] for k in ("partials", "heavy_rows", "heavy_partials", "sorted_segments", "sorted_partials", "light_tiles", "ms_total")})
a = res.to_host(); b = m.to_host()
print("rowptr equal", np.array_equal(a[0], b[0]), "first diff row", int(np.argmax(a[0] != b[0])) if not np.array_equal(a[0], b[0]) else -1)
r = int(np.argmax(a[0] != b[0])) - 1
print("row", r, "len expected", a[0][r + 1] - a[0][r], "got", b[0][r + 1] - b[0][r])
