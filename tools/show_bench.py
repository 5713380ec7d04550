"""Print the figures of a bench.py JSON line that matter when comparing two builds."""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"value {d['value'] / 1e9:.2f} G nnz/s   {d['ms_per_step']:.1f} ms/step   n_gpus {d['n_gpus']}")
r = d.get("roofline", {})
print(f"roofline: {r.get('kernel')} {r.get('achieved', 0):.0f} GB/s = {r.get('frac', 0):.3f} of peak")
for k, v in r.get("kernels", {}).items():
    print(f"   {k:22s} {v['ms_per_launch']:7.2f} ms x {v['launches_per_step']}  {v['GBps']:7.0f} GB/s  ({v['GBps'] / 8000:.3f})")
if "whole_product" in d:
    print(f"whole product: {d['whole_product']['GBps']:.0f} GB/s algorithmic = {d['whole_product']['frac_of_peak']:.3f} of peak")
print("phases:", {k: round(v, 1) for k, v in d.get("phases_ms", {}).items()})
if "steps_ms" in d:
    print("steps total:", d["steps_ms"]["total"])
if "cpu_baseline" in d:
    print("cpu:", f"{d['cpu_baseline']['value'] / 1e6:.1f} M nnz/s ({d['cpu_baseline']['kind']})", " speedup", round(d.get("speedup_vs_cpu", 0)))
    print("slab parity:", d.get("slab_parity"))
for k, v in d.get("extra_workloads", {}).items():
    ks = {kk: (round(vv["ms_per_launch"], 2), round(vv["GBps"])) for kk, vv in v["kernels"].items()}
    print(f"extra {k}: {v['ms_per_step']:.2f} ms  {v['value'] / 1e9:.2f} G nnz/s  whole {v['whole_product_frac_of_peak']:.3f}  {ks}")
for k, v in d.get("decompositions", {}).items():
    print(f"decomposition {k}: {v['ms_per_step']:.1f} ms  {v['value'] / 1e9:.2f} G nnz/s  {v.get('rank0_phases_ms')}")
