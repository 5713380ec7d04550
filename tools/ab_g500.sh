for cfg in "--rmat g500 --scale 20 --stream-output" "--rmat g500 --scale 22 --stream-output" "--rmat g500 --scale 18 --edge-factor 64"; do
    timeout -k 10 300 python bench.py $cfg --cpu-baseline 0 --extras 0 --ingest 0 --steps 2 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED $cfg"; tail -3 gpurun_out/ab.err; continue; }
    python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(d['ms_per_step'],1), 'serial', d.get('ms_per_step_plans_in_line'), {k:(round(v['ms_per_launch'],2), round(v.get('ms_per_launch_beside',0),2), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})
PY
done
