#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_omega_lds.py tests/test_gpu_admm.py tests/test_gpu_batch_isolation.py tests/test_gpu_ext.py -x -q > $O/pytest_q.txt 2>&1
tail -3 $O/pytest_q.txt
for i in 1 2 3; do
python bench.py --workload ggl_K256_p64 --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ggl_K256_p64', round(d['value'],1), d['ms_per_step'], d.get('phases_ms_per_step'))"
done
python tools/time_ctx.py 2>&1 | tail -12
