#!/usr/bin/env python3
"""Where one product's wall time goes BETWEEN kernels: reads a rocprofv3 kernel trace (csv), takes the last product
(from the last symbolic sort to the end), and lists the idle gaps on the device with the kernels either side.
usage: tools/gap_report.py KERNEL_TRACE.csv [MEMORY_COPY_TRACE.csv] [--min-us 30]"""
import csv
import sys


def main():
    argv = sys.argv[1:]
    min_us = 30.0
    if "--min-us" in argv:
        i = argv.index("--min-us")
        min_us = float(argv[i + 1])
        del argv[i:i + 2]
    args = argv
    ev = []
    with open(args[0]) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:90]))
    if len(args) > 1:
        with open(args[1]) as f:
            for r in csv.DictReader(f):
                ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
    ev.sort()
    # the last product starts at the last rs_hist_kernel that follows a non-osp kernel or a long gap
    starts = [i for i, e in enumerate(ev) if "sym_chunk_len_kernel" in e[2]]
    if not starts:
        print("no product found")
        return
    i0 = starts[-1]
    while i0 > 0 and "osp::" in ev[i0 - 1][2] and ev[i0][0] - ev[i0 - 1][1] < 2_000_000:
        i0 -= 1
    i1 = len(ev) - 1
    while i1 > i0 and "osp::" not in ev[i1][2]:
        i1 -= 1
    prod = ev[i0:i1 + 1]
    wall = (prod[-1][1] - prod[0][0]) / 1e6
    busy = 0.0
    end = prod[0][0]
    gaps = []
    for s, e, name in prod:
        if s > end:
            gaps.append(((s - end) / 1e3, name))
        if e > end:
            busy += (e - max(s, end)) / 1e6
            end = e
    print(f"last product: {len(prod)} device operations, wall {wall:.2f} ms, device busy {busy:.2f} ms, idle {wall - busy:.2f} ms")
    big = [g for g in gaps if g[0] >= min_us]
    print(f"gaps >= {min_us:.0f} us: {len(big)} totalling {sum(g[0] for g in big) / 1e3:.2f} ms; smaller gaps: {len(gaps) - len(big)} totalling "
          f"{sum(g[0] for g in gaps if g[0] < min_us) / 1e3:.2f} ms")
    end = prod[0][0]
    last = None
    for s, e, name in prod:
        if s > end and (s - end) / 1e3 >= min_us:
            print(f"  {(s - end) / 1e3:9.1f} us  after {last[:60]:60s} before {name[:60]}")
        if e > end:
            end = e
            last = name


if __name__ == "__main__":
    main()
