#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_omega_lds.py tests/test_gpu_batch_isolation.py tests/test_gpu_selection.py -x -q > $O/pytest_i.txt 2>&1
tail -5 $O/pytest_i.txt
python -m pytest tests/test_gpu_admm.py tests/test_gpu_latent_rank.py -x -q -k "batch or grid or sgl or block" > $O/pytest_i2.txt 2>&1
tail -3 $O/pytest_i2.txt
python tools/time_batch.py --p 64 --points 100
python tools/time_batch.py --p 50 --points 20
python tools/time_batch.py --p 30 --points 20
python tools/bench_grid.py --p 50 --points 20 --no-sequential 2>&1 | grep "^{" | cut -c1-200
python tools/bench_grid.py --p 64 --points 100 --no-sequential 2>&1 | grep "^{" | cut -c1-200
