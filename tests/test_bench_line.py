"""The bench line the driver reads (bench.py's contract), checked on the line committed under profiles/: the keys the contract
names, and the arithmetic between them.  CPU-only: it reads a JSON file."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_default_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    if not paths:
        pytest.skip("no committed bench line")
    with open(paths[-1]) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
    return paths[-1], json.loads(lines[-1])


def test_committed_bench_line_keeps_the_contract():
    path, d = _newest_default_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (path, k)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None   # BASELINE.md publishes no number
    assert d["dtype"] in ("f64", "f32") and d["data"] == "synthetic" and d["unit"] == "nnz/s"
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg
    # value = output entries of one product over the time of one step
    assert d["value"] == pytest.approx(cfg["nnz_c"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.0 < r["frac"] < 1.0
    # the dominant kernel's launches fit inside the step they are part of
    k = r["kernels"][r["kernel"]]
    assert k["ms_per_launch"] * k["launches_per_step"] <= d["ms_per_step"]
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (k["ms_per_launch"] * 1e-3) / 1e9, rel=1e-6)
    # counters, when present, are per launch like `achieved` and belong to the library that ran
    if r["traffic"] is not None:
        assert r["traffic"] > 0 and r.get("traffic_stale") in (False, None)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == "nnz/s" and c["value"] > 0
    # both full-size parity checks of the run passed
    assert d["slab_parity"]["status"] == "ok" and d["row_slab_parity"]["status"] == "ok"
