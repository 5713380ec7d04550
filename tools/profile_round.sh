#!/bin/bash
# Collects what profiles/ holds for one round on the GPU box: bench lines, rocprofv3 kernel statistics and the two PMC
# passes (FETCH_SIZE, WRITE_SIZE -- separate runs, never combined with other trace domains).
# usage: tools/profile_round.sh OUTDIR     (run from the repository root; results land in OUTDIR)
set -o pipefail
R=$(pwd)
O=$R/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 $R/bench.py --rmat uniform --cpu-baseline 0 --extras 0 > $O/bench_uniform.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype f32 --cpu-baseline 0 --extras 0 > $O/bench_mild_f32.json 2>/dev/null || exit 1
python3 $R/bench.py --workload webgoogle --cpu-baseline 0 --steps 20 --warmup 3 > $O/bench_webgoogle.json 2>/dev/null || exit 1
python3 $R/bench.py --rmat g500 --scale 20 --stream-output --cpu-baseline 0 --steps 2 --warmup 1 > $O/bench_g500_20_streamed.json 2>/dev/null || exit 1
for w in mild uniform; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 $R/bench.py --rmat $w --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 --ingest 0 > $O/ks_$w.json 2> $O/ks_$w.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_webgoogle -- python3 $R/bench.py --workload webgoogle --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 --ingest 0 > $O/ks_webgoogle.json 2> $O/ks_webgoogle.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_g500 -- python3 $R/bench.py --rmat g500 --scale 20 --stream-output --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 --ingest 0 > $O/ks_g500.json 2> $O/ks_g500.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --extras 0 --ingest 0 > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
done
# keep only the small summaries (the traces themselves are large)
for w in mild uniform g500 webgoogle; do
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -n 1)
  cp "$f" $O/ks_$w.kernel_stats.csv
  rm -rf $O/ks_$w
done
for c in FETCH_SIZE WRITE_SIZE; do
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" $O/pmc_$c.csv <<'PY'
import csv, sys
# per-kernel sums only: the per-dispatch file is tens of MB
from collections import defaultdict
tot = defaultdict(float); calls = defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        tot[row["Kernel_Name"]] += float(row["Counter_Value"]); calls[row["Kernel_Name"]] += 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Kernel_Name", "Counter_Value", "Calls"])
    for k in tot: w.writerow([k, tot[k], calls[k]])
PY
  rm -rf $O/pmc_$c
done
ls -la $O
