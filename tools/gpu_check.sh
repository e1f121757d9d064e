#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test suite + headline bench line.   tools/gpu_check.sh <tag> [pytest args]
set -u
TAG=${1:-chk}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -q -x --durations=15 "$@" ) > $O/pytest_gpu.txt 2>&1
tail -30 $O/pytest_gpu.txt
python bench.py --no-cpu-baseline > $O/bench.log 2>&1
grep "^{" $O/bench.log | head -c 1500
