#!/usr/bin/env python3
"""The last product of a rocprofv3 kernel trace as a timeline: every device operation with its duration and the idle gap
before it.  usage: tools/timeline.py KERNEL_TRACE.csv"""
import csv
import sys


def main():
    ev = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("osp::", "")[:70]))
    ev.sort()
    starts = [i for i, e in enumerate(ev) if "count_partials_kernel" in e[2]]
    if not starts:
        print("no product found")
        return
    i0 = starts[-1]
    prod = [e for e in ev[i0:] if e[0] - ev[i0][0] < 50_000_000]
    t0 = prod[0][0]
    end = t0
    tot = gaps = 0.0
    for s, e, name in prod:
        gap = max(0, s - end) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} us  +{gap:7.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
        tot += (e - s) / 1e3
        gaps += gap
        end = max(end, e)
    print(f"{len(prod)} operations, kernels {tot / 1e3:.3f} ms, gaps {gaps / 1e3:.3f} ms, wall {(end - t0) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
