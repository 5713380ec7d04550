#!/usr/bin/env python3
"""SQ counters of the main kernels -> a small markdown table for profiles/ (what a kernel is busy with, what it waits for).

usage: summarize_sq.py pmc_SQA.csv pmc_SQB.csv out.md "title"
Inputs: the per-kernel sums tools/profile_round.sh writes (Kernel_Name, Counter_Name, Counter_Value, Calls).
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md): the shares below
are of the waves' lifetime.  SQ_BUSY_CYCLES is summed over the SQs; shares of it say how busy a pipe is while the kernel runs."""
import csv
import re
import sys
from collections import defaultdict

KEEP = ("merge_tiles_kernel", "multiply_kernel", "direct_plan_kernel", "split_scatter_kernel", "dense_segment_kernel")


def short(name):
    m = re.search(r"osp::([A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    if not m:
        return None
    base = m.group(1)
    if base == "merge_tiles_kernel" and not re.search(r"<[^,]+, \d+, 0,", m.group(2) or ""):
        return base + " (in-place)"
    return base


def load(path, acc, calls):
    with open(path) as f:
        for r in csv.DictReader(f):
            s = short(r["Kernel_Name"])
            if s is None:
                continue
            acc[s][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[s] = max(calls[s], int(r.get("Calls") or 1))


def main(a, b, out, title):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for p in (a, b):
        try:
            load(p, acc, calls)
        except OSError:
            pass
    names = [k for k in KEEP if k in acc]
    with open(out, "w") as f:
        f.write(f"# {title}\n\nTwo `rocprofv3 --pmc ... --kernel-trace` passes of eight SQ counters each over one product "
                "(`bench.py --steps 1 --warmup 0`), summed per kernel.  Shares of the waves' lifetime (SQ_WAVE_CYCLES): parked = "
                "SQ_WAIT_ANY (s_waitcnt / barrier), issue stall = SQ_WAIT_INST_ANY, issuing = SQ_ACTIVE_INST_ANY.  VALU busy = "
                "SQ_ACTIVE_INST_VALU / (8 x SQ_BUSY_CYCLES): quad-cycles x 4, summed over 256 CUs x 4 SIMDs, over the kernel's busy cycles "
                "(SQ_BUSY_CYCLES is summed over the chip's 32 SQ instances); LDS busy = SQ_LDS_IDX_ACTIVE / (8 x SQ_BUSY_CYCLES): the LDS "
                "array's active cycles per CU over the same time.  Bank conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  (Own "
                "derivation: ROCm 7.2 ships no gfx950 formulas.)\n\n")
        f.write("| kernel | launches | waves parked | issue stall | issuing | VALU busy | LDS busy | LDS bank-conflict share | VALU / SALU / LDS / VMEM-rd / VMEM-wr instructions per wave-cycle x1000 |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|---:|---|\n")
        for k in names:
            c = acc[k]
            wc = c.get("SQ_WAVE_CYCLES", 0) or float("nan")
            busy = c.get("SQ_BUSY_CYCLES", 0) or float("nan")
            def pct(x, d):
                return f"{100.0 * x / d:.0f} %" if d == d and d else "-"
            lds_act = c.get("SQ_LDS_IDX_ACTIVE", 0)
            f.write(f"| `{k}` | {calls[k]} | {pct(c.get('SQ_WAIT_ANY', 0), wc)} | {pct(c.get('SQ_WAIT_INST_ANY', 0), wc)} | "
                    f"{pct(c.get('SQ_ACTIVE_INST_ANY', 0), wc)} | {pct(c.get('SQ_ACTIVE_INST_VALU', 0), 8 * busy)} | "
                    f"{pct(c.get('SQ_LDS_IDX_ACTIVE', 0), 8 * busy)} | {pct(c.get('SQ_LDS_BANK_CONFLICT', 0), lds_act) if lds_act else '-'} | "
                    + " / ".join(f"{1000.0 * c.get(n, 0) / wc:.1f}" if wc == wc else "-" for n in
                                 ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) + " |\n")
        f.write("\nRaw sums:\n\n| kernel | " + " | ".join(sorted({n for k in names for n in acc[k]})) + " |\n")
        cols = sorted({n for k in names for n in acc[k]})
        f.write("|---|" + "---:|" * len(cols) + "\n")
        for k in names:
            f.write(f"| `{k}` | " + " | ".join(f"{acc[k].get(n, 0):.4g}" for n in cols) + " |\n")


if __name__ == "__main__":
    main(*sys.argv[1:5])
