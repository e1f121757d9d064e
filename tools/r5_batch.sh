#!/bin/bash
# round 5: the batch drivers' host loop in C (ggl_sgl_batch_run / ggl_mgl_batch_run), the packed K-sharded exchange
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_batch_isolation.py tests/test_gpu_selection.py tests/test_gpu_latent_rank.py tests/test_gpu_multirank.py -x -q > $O/pytest_batch.txt 2>&1
tail -3 $O/pytest_batch.txt
python -m pytest tests/test_gpu_admm.py tests/test_gpu_ext.py -x -q > $O/pytest_batch2.txt 2>&1
tail -3 $O/pytest_batch2.txt
for a in "--p 50 --points 20" "--p 64 --points 100" "--p 1000 --points 20"; do
  python tools/bench_grid.py $a --no-sequential 2>&1 | grep "^{" > "$O/grid_$(echo $a | tr -d ' -').json"
done
python tools/bench_mgl_grid.py 2>&1 | grep "^{" > $O/mgl_grid_8x1_K4_p500.json
python tools/bench_mgl_grid.py --reg FGL --K 6 --p 300 --l1 4 --l2 3 2>&1 | grep "^{" > $O/mgl_grid_4x3_fgl_K6_p300.json
for w in ggl_K4_p500 ggl_K8_p500 ggl_K16_p500; do
  python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_$w.json
  GGL_BENCH_FORCE_DIST=1 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --comm capi 2>&1 | grep "^{" > $O/workload_${w}_sharded_1rank_rccl_capi.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/grid_*.json")+glob.glob("$O/mgl_grid*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], {k:d[k] for k in d if k in ('batched_solve_s','batched_iterations_max','single_grid_search_total_s','solve_s','grid_search_total_s')})
for f in sorted(glob.glob("$O/workload_ggl_K*_p500*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), d['ms_per_step'], d.get('phases_ms'))
PY
