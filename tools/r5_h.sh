#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_batch_isolation.py tests/test_gpu_selection.py tests/test_gpu_ops.py -x -q -k "batch or int8 or selection or grid" > $O/pytest_h.txt 2>&1
tail -3 $O/pytest_h.txt
python -m pytest tests/test_gpu_admm.py tests/test_gpu_ext.py tests/test_gpu_latent_rank.py -x -q -k "batch or grid or sgl or ext" > $O/pytest_h2.txt 2>&1
tail -3 $O/pytest_h2.txt
python tools/time_batch.py --p 64 --points 100
python tools/time_batch.py --p 50 --points 20
python tools/bench_grid.py --p 50 --points 20 --no-sequential 2>&1 | grep "^{" | cut -c1-200
python tools/bench_grid.py --p 64 --points 100 --no-sequential 2>&1 | grep "^{" | cut -c1-200
python tools/bench_mgl_grid.py 2>&1 | grep "^{" | cut -c1-300
python bench.py --workload ggl_K20_p200 --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', round(d['value'],1), d['ms_per_step'], d.get('phases_ms_per_step'), d['newton_schulz'])"
