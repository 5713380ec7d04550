for a in "--rmat uniform --scale 12 --edge-factor 1000" "--rmat uniform --scale 18 --edge-factor 200" "--rmat g500 --scale 14 --edge-factor 512" "--rmat mild --scale 20 --edge-factor 32" "--rmat uniform --scale 24 --edge-factor 6"; do
  python3 bench.py $a --cpu-baseline 0 --extras 0 --ingest 0 --steps 2 --warmup 1 2>gpurun_out/pr.err > gpurun_out/pr.json || { echo "FAILED $a"; tail -3 gpurun_out/pr.err; continue; }
  python3 - "$a" <<PY
import json,sys
d=json.loads([l for l in open("gpurun_out/pr.json") if l.strip().startswith("{")][0])
P=d["config"]["partials"]; C=d["config"]["nnz_c"]; ms=d["ms_per_step"]
k=d["roofline"]["kernels"]
print(f"{sys.argv[1]:55s} P={P/1e9:7.2f}G nnzC={C/1e9:6.2f}G {ms:8.1f} ms  {ms*1e9/P:6.1f} ps/product  panels {d['panels']}  " + " ".join(f"{n.split('_kernel')[0]}={v['ms_per_launch']*v['launches_per_step']:.1f}" for n,v in k.items()))
PY
done
