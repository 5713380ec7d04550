#!/usr/bin/env python3
"""Trim a rocprofv3 `--kernel-trace --stats` kernel_stats.csv to this library's kernels (osp::*),
shorten the names and write a small CSV + markdown table for profiles/."""
import csv
import re
import sys


def short(name):
    m = re.search(r"osp::([A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    if not m:
        return None
    targs = m.group(2) or ""
    targs = targs.replace("unsigned int", "u32").replace("unsigned long", "u64").replace("osp::", "")
    return m.group(1) + targs


def main(src, dst_prefix, title):
    rows = []
    other_ns = 0
    with open(src) as f:
        for r in csv.DictReader(f):
            s = short(r["Name"])
            if s is None:
                other_ns += int(r["TotalDurationNs"])
                continue
            rows.append((s, int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), int(r["MinNs"]), int(r["MaxNs"])))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    with open(dst_prefix + ".csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct_of_osp"])
        for r in rows:
            w.writerow(list(r) + [f"{100.0 * r[2] / tot:.2f}"])
    with open(dst_prefix + ".md", "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats`; only this library's kernels (osp::*) are listed; "
                f"other kernels in the process (torch input generation etc.) total {other_ns / 1e6:.2f} ms.\n\n")
        f.write("| kernel | calls | avg ms | total ms | % of osp time |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            f.write(f"| `{r[0]}` | {r[1]} | {r[3] / 1e6:.4f} | {r[2] / 1e6:.3f} | {100.0 * r[2] / tot:.1f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
