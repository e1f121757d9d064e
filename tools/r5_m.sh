#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/time_batch.py --p 64 --points 100 2>&1 | grep -v amdgpu
python tools/bench_grid.py --p 64 --points 100 --no-sequential 2>&1 | grep "^{" | cut -c1-420
