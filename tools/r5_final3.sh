#!/bin/bash
# the round's last GPU call: suite + bench on the final tree, the download measurement
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/r5_final.sh
timeout 400 python tools/time_download.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/download.txt
tail -6 gpurun_out/r5/download.txt | cut -c1-200
python tools/bench_grid.py 2>&1 | grep "^{" > gpurun_out/r5/workload_sgl_grid_p1000_L20.json
python -c "
import json; d=json.load(open('gpurun_out/r5/workload_sgl_grid_p1000_L20.json')); print('C2', d['batched_solve_s'], d['single_grid_search_total_s'])"
