#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python tools/bench_omega_lds.py 256:64 256:50 > $O/omega_lds_kernel.txt 2>&1
grep "K=" $O/omega_lds_kernel.txt | grep "2e-12" | head -8
( time python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
