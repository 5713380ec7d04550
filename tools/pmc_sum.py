"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter)."""
import csv
import sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k] += 1
names = sorted({c for k in tot for c in tot[k]})
keep = [k for k in tot if "osp::" in k]
keep.sort(key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get(names[0], 0)))
print("kernel," + ",".join(names))
for k in keep[:12]:
    print(k[:60] + "," + ",".join(f"{tot[k].get(c, 0):.4g}" for c in names))
