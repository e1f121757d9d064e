#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_omega_lds.py tests/test_gpu_batch_isolation.py -x -q > $O/pytest_d.txt 2>&1
tail -3 $O/pytest_d.txt
python -m pytest tests/test_gpu_admm.py -x -q -k "pipelined or sharded or speculative" > $O/pytest_d2.txt 2>&1
tail -3 $O/pytest_d2.txt
python tools/time_batch.py --p 1000 --points 20 --compact 1
for i in 1 2; do
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline fused_w=1', round(d['value'],1), d['ms_per_step'], d.get('phases_ms'))"
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --opt fused_w=0 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline fused_w=0', round(d['value'],1), d['ms_per_step'], d.get('phases_ms'))"
done
for w in ggl_K4_p500 ggl_K20_p200 ggl_K8_p500; do
for fw in 1 0; do
  python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --opt fused_w=$fw 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w fused_w=$fw', round(d['value'],1), d['ms_per_step'], d.get('phases_ms'))"
done; done
