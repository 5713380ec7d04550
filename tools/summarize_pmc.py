#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel (this library's osp::* kernels).

usage: summarize_pmc.py FETCH_SIZE.csv WRITE_SIZE.csv bench.json out_prefix "title"
FETCH_SIZE / WRITE_SIZE are reported in KiB (bytes = value * 1024).  gfx950 correction from
MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies wide coalesced streaming reads at half their size, so the
table shows the raw figure AND the doubled one; WRITE_SIZE is exact for streaming stores.
Writes out_prefix.md and out_prefix.json (per-kernel bytes per launch, used by bench.py's roofline.traffic)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"osp::([A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    if not m:
        return None
    base = m.group(1)
    if base == "merge_tiles_kernel":
        # the tile kernel has two instantiations per value type: the chained one (mode 0, what bench.py's roofline is
        # about) and the big in-place tiles for over-long segments (mode 32)
        return base if re.search(r"<[^,]+, \d+, 0,", m.group(2) or "") else base + " (in-place)"
    return base


def load(path):
    tot = defaultdict(float)
    calls = defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            s = short(r["Kernel_Name"])
            if s is None:
                continue
            tot[s] += float(r["Counter_Value"]) * 1024.0
            calls[s] += int(r.get("Calls") or 1)  # (tools/profile_round.sh stores per-kernel sums with a call count)
    return tot, calls


def main(fetch_csv, write_csv, bench_json, out_prefix, title):
    fetch, calls = load(fetch_csv)
    write, _ = load(write_csv)
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    alg = {k: v["algorithmic_bytes_per_launch"] for k, v in bench["roofline"]["kernels"].items()}
    names = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0)))
    # library_id: which kernels these counters belong to (bench.py compares it with the library it runs: roofline.traffic_stale)
    out = {"workload": bench["config"]["workload"], "dtype": bench["dtype"], "source": out_prefix + ".md", "library_id": bench.get("library_id"),
           "kernels": {}}
    with open(out_prefix + ".md", "w") as f:
        f.write(f"# {title}\n\nTwo separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace`, `--pmc WRITE_SIZE --kernel-trace`); "
                "counter unit KiB.  gfx950: FETCH_SIZE counts wide coalesced reads at half their size "
                "(MI355X_MICROARCH.md, HBM) -- `fetch x2` is the corrected figure for streaming reads; WRITE_SIZE is exact.\n\n"
                "| kernel | launches | FETCH_SIZE / launch | fetch x2 / launch | WRITE_SIZE / launch | algorithmic bytes / launch |\n"
                "|---|---:|---:|---:|---:|---:|\n")
        for k in names:
            c = max(calls.get(k, 1), 1)
            fr, wr = fetch.get(k, 0.0) / c, write.get(k, 0.0) / c
            a = alg.get(k)
            f.write(f"| `{k}` | {c} | {fr / 1e9:.3f} GB | {2 * fr / 1e9:.3f} GB | {wr / 1e9:.3f} GB | "
                    f"{(f'{a / 1e9:.3f} GB' if a else '-')} |\n")
            if a:
                out["kernels"][k] = {"fetch_raw": fr, "fetch_x2": 2 * fr, "write": wr, "traffic": 2 * fr + wr, "algorithmic": a}
    with open(out_prefix + ".json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:6])
