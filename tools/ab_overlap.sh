# A/B of planning panel p+1 beside the multiply of panel p (OSP_PLAN_OVERLAP=1/0 for the whole process); usage: bash tools/ab_overlap.sh [values] [configs...]
V=${1:-"1 0"}
shift
if [ $# -eq 0 ]; then set -- "--rmat mild --scale 22" "--rmat g500 --scale 20 --stream-output" "--rmat mild --scale 22 --dtype f32" "--rmat g500 --scale 22 --stream-output"; fi
for cfg in "$@"; do
  for h in $V; do
    OSP_PLAN_OVERLAP=$h timeout -k 10 300 python bench.py $cfg --cpu-baseline 0 --extras 0 --ingest 0 --steps 3 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED $cfg overlap=$h"; tail -3 gpurun_out/ab.err; continue; }
    python - "$cfg" $h <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'overlap='+sys.argv[2], round(d['ms_per_step'],1), 'steps', d['steps_ms']['total'], 'in-line steps of the same process', round(d.get('ms_per_step_plans_in_line') or 0,1), 'panels', d.get('panels'),
      {k:(round(v['ms_per_launch'],2), round(v.get('ms_per_launch_beside',0),2), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})
PY
  done
done
