#!/bin/bash
# round 5: the batch drivers' host loop in C (ggl_sgl_batch_run / ggl_mgl_batch_run) -- tests and the small grids
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_batch_isolation.py tests/test_gpu_selection.py tests/test_gpu_latent_rank.py -x -q -k "not dispatch" > $O/pytest_batch.txt 2>&1
tail -3 $O/pytest_batch.txt
python -m pytest tests/test_gpu_admm.py -x -q -k "batch or grid or sgl" > $O/pytest_batch2.txt 2>&1
tail -3 $O/pytest_batch2.txt
for a in "--p 50 --points 20" "--p 64 --points 100" "--p 1000 --points 20"; do
  python tools/bench_grid.py $a --no-sequential 2>&1 | grep "^{" > "$O/grid_$(echo $a | tr -d ' -').json"
done
python tools/bench_mgl_grid.py 2>&1 | grep "^{" > $O/mgl_grid_8x1_K4_p500.json
python tools/bench_mgl_grid.py --reg FGL --K 6 --p 300 --l1 4 --l2 3 2>&1 | grep "^{" > $O/mgl_grid_4x3_fgl_K6_p300.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/grid_*.json")+glob.glob("$O/mgl_grid*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], {k:d[k] for k in d if k in ('batched_solve_s','batched_iterations_max','single_grid_search_total_s','solve_s','grid_search_total_s','uncompacted')})
PY
