#!/bin/bash
# Collects what profiles/ holds for one round on the GPU box: bench lines, rocprofv3 kernel statistics and the two PMC
# passes (FETCH_SIZE, WRITE_SIZE -- separate runs, never combined with other trace domains).
# usage: tools/profile_round.sh OUTDIR     (run from the repository root; results land in OUTDIR)
set -o pipefail
R=$(pwd)
O=$R/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# PART=a: the bench lines; PART=b: kernel statistics; PART=c: the counter passes; unset: everything (about 20 minutes of box
# time: more than one gpurun call allows)
if [ -z "$PART" ] || [ "$PART" = a ]; then
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python3 $R/bench.py --rmat uniform --cpu-baseline 0 --extras 0 > $O/bench_uniform.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype f32 --cpu-baseline 0 --extras 0 > $O/bench_mild_f32.json 2>/dev/null || exit 1
python3 $R/bench.py --workload webgoogle --cpu-baseline 0 --steps 20 --warmup 3 > $O/bench_webgoogle.json 2>/dev/null || exit 1
python3 $R/bench.py --rmat g500 --scale 20 --stream-output --cpu-baseline 0 --steps 2 --warmup 1 > $O/bench_g500_20_streamed.json 2>/dev/null || exit 1
fi
if [ -z "$PART" ] || [ "$PART" = b ]; then
for w in mild uniform; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 $R/bench.py --rmat $w --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 --ingest 0 > $O/ks_$w.json 2> $O/ks_$w.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_webgoogle -- python3 $R/bench.py --workload webgoogle --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 --ingest 0 > $O/ks_webgoogle.json 2> $O/ks_webgoogle.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_g500 -- python3 $R/bench.py --rmat g500 --scale 20 --stream-output --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 --ingest 0 > $O/ks_g500.json 2> $O/ks_g500.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_g500_22 -- python3 $R/bench.py --rmat g500 --scale 22 --stream-output --steps 1 --warmup 1 --cpu-baseline 0 --extras 0 --ingest 0 > $O/ks_g500_22.json 2> $O/ks_g500_22.err || exit 1
fi
if [ -z "$PART" ] || [ "$PART" = c ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --extras 0 --ingest 0 > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
done
# the SQ counters of the three main kernels (what they are busy with, what they wait for): two passes of at most 8 counters
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $O/pmc_SQA -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --extras 0 --ingest 0 > $O/pmc_SQA.json 2> $O/pmc_SQA.err || echo "SQ pass A failed" >> $O/notes.txt
rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $O/pmc_SQB -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0 --extras 0 --ingest 0 > $O/pmc_SQB.json 2> $O/pmc_SQB.err || echo "SQ pass B failed" >> $O/notes.txt
fi
# keep only the small summaries (the traces themselves are large)
for w in mild uniform g500 webgoogle g500_22; do
  [ -d $O/ks_$w ] || continue
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -n 1)
  cp "$f" $O/ks_$w.kernel_stats.csv
  rm -rf $O/ks_$w
done
for c in FETCH_SIZE WRITE_SIZE SQA SQB; do
  [ -d $O/pmc_$c ] || continue
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] || continue
  python3 - "$f" $O/pmc_$c.csv <<'PY'
import csv, sys
# per-kernel sums only: the per-dispatch file is tens of MB
from collections import defaultdict
tot = defaultdict(float); calls = defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        key = (row["Kernel_Name"], row.get("Counter_Name", ""))
        tot[key] += float(row["Counter_Value"]); calls[key] += 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Calls"])
    for k in tot: w.writerow([k[0], k[1], tot[k], calls[k]])
PY
  rm -rf $O/pmc_$c
done
ls -la $O
