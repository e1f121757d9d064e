cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
python tools/bench_grid.py 2>&1 | grep "^{" > gpurun_out/r5/workload_sgl_grid_p1000_L20.json
python tools/bench_mgl_grid.py 2>&1 | grep "^{" > gpurun_out/r5/workload_mgl_grid_8x1_K4_p500.json
python tools/bench_mgl_grid.py --reg FGL --K 6 --p 300 --l1 4 --l2 3 2>&1 | grep "^{" > gpurun_out/r5/workload_mgl_grid_4x3_fgl_K6_p300.json
python - <<'PY'
import json
for f in ("workload_sgl_grid_p1000_L20","workload_mgl_grid_8x1_K4_p500","workload_mgl_grid_4x3_fgl_K6_p300"):
    d=json.load(open("gpurun_out/r5/"+f+".json")); print(f, {k: round(v,4) for k,v in d.items() if isinstance(v,float)})
PY
