# A/B of the hub rows (OSP_HUB=1/0) on regimes beside the bench's defaults; usage: bash tools/ab_hub.sh [hub values]
H=${1:-"1 0"}
for cfg in "--rmat g500 --scale 18 --edge-factor 64" "--rmat g500 --scale 17 --edge-factor 64" "--rmat mild --scale 19 --edge-factor 64" "--rmat mild --scale 18 --edge-factor 32" "--rmat g500 --scale 16 --edge-factor 128 --dtype f32" "--rmat g500 --scale 20 --stream-output" "--rmat mild --scale 20 --edge-factor 32"; do
  for h in $H; do
    OSP_HUB=$h timeout -k 10 200 python bench.py $cfg --cpu-baseline 0 --extras 0 --ingest 0 --steps 2 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED $cfg hub=$h"; tail -3 gpurun_out/ab.err; continue; }
    python - "$cfg" $h <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
hp, hc = d.get('long_row_partials_hub') or 0, d.get('hub_cells') or 0
print(sys.argv[1], 'hub='+sys.argv[2], round(d['ms_per_step'],1), 'hub products', hp, 'of', d['config']['partials'], 'records per run', round(hp / hc, 2) if hc else '-', {k:(round(v['ms_per_launch'],2), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})
PY
  done
done
