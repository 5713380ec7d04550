import csv, sys, glob
d = sys.argv[1]
api = []
for f in glob.glob(d + "/*hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
api.sort()
ker = []
for f in glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("osp::", "")[:50]))
ker.sort()
starts = [i for i, e in enumerate(ker) if "count_partials_kernel" in e[2]]
t0 = ker[starts[-1]][0]
# host API calls from 100 us before the last product's first kernel, for 3 ms
print("host API calls longer than 8 us, and all syncs (relative to the product's first kernel):")
for s, e, fn in api:
    if s < t0 - 200_000 or s > t0 + 3_000_000: continue
    if e - s > 8000 or "Synchronize" in fn:
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {fn}")
