# A/B of planning panel p+1 beside the multiply of panel p (OSP_PLAN_OVERLAP=1/0); usage: bash tools/ab_overlap.sh [values]
V=${1:-"1 0"}
for cfg in "--rmat mild --scale 22" "--rmat g500 --scale 20 --stream-output" "--rmat mild --scale 22 --dtype f32" "--rmat g500 --scale 22 --stream-output"; do
  for h in $V; do
    OSP_PLAN_OVERLAP=$h timeout -k 10 300 python bench.py $cfg --cpu-baseline 0 --extras 0 --ingest 0 --steps 2 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED $cfg overlap=$h"; tail -3 gpurun_out/ab.err; continue; }
    python - "$cfg" $h <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'overlap='+sys.argv[2], round(d['ms_per_step'],1), 'panels', d.get('panels'), {k:(round(v['ms_per_launch'],2), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})
PY
  done
done
