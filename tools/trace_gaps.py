"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: where the host makes the device wait."""
import csv
import sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
# the library's products only: from the first to the last osp:: kernel of the LAST product (marked by sym_chunk_len_kernel)
starts = [i for i, r in enumerate(rows) if "sym_chunk_len_kernel" in r[2]]
lo = starts[-1]
hi = max(i for i, r in enumerate(rows) if "osp::" in r[2])
seg = rows[lo:hi + 1]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps = defaultdict(lambda: [0, 0])
end = seg[0][1]
for s, e, n in seg[1:]:
    if s > end:
        g = gaps[(prev_name, n)] if False else None
    end = max(end, e)
end = seg[0][1]
prev = seg[0][2]
tot_gap = 0
for s, e, n in seg[1:]:
    if s > end:
        gaps[prev + "  ->  " + n][0] += s - end
        gaps[prev + "  ->  " + n][1] += 1
        tot_gap += s - end
    if e > end:
        end = e
        prev = n
print(f"last product: span {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms, idle {tot_gap / 1e6:.2f} ms, {len(seg)} launches")
for k, (ns, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {ns / 1e3:9.1f} us in {c:3d} gaps   {k[:150]}")
