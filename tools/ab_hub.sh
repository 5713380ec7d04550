# A/B of the hub rows on regimes beside the bench's defaults: default thresholds / forced (every panel with stretch rows) / off;
# usage: bash tools/ab_hub.sh ["default forced off"]
H=${1:-"default forced off"}
for cfg in "--rmat g500 --scale 18 --edge-factor 64" "--rmat mild --scale 19 --edge-factor 64" "--rmat mild --scale 18 --edge-factor 32" "--rmat g500 --scale 16 --edge-factor 128 --dtype f32" "--rmat g500 --scale 20 --stream-output" "--rmat mild --scale 20 --edge-factor 32" "--rmat mild --scale 22"; do
  for h in $H; do
    case $h in
      default) E="";;
      forced) E="OSP_HUB_MIN_SHARE=0 OSP_HUB_MIN_RUN=0";;
      off) E="OSP_HUB=0";;
    esac
    env $E timeout -k 10 200 python bench.py $cfg --cpu-baseline 0 --extras 0 --ingest 0 --steps 2 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED $cfg hub=$h"; tail -3 gpurun_out/ab.err; continue; }
    python - "$cfg" $h <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
hp, hc = d.get('long_row_partials_hub') or 0, d.get('hub_cells') or 0
print(sys.argv[1], 'hub='+sys.argv[2], round(d['ms_per_step'],1), 'hub products', hp, 'of', d['config']['partials'], 'records per run', round(hp / hc, 2) if hc else '-', {k:(round(v['ms_per_launch'],2), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})
PY
  done
done
