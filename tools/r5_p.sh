#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
cd $R
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-exact-region 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline default', round(d['value'],1), d['ms_per_step'], d['roofline']['frac'])"
python bench.py --no-cpu-baseline --no-exact-region --opt bound_side=0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline bound_side=0', round(d['value'],1), d['ms_per_step'], d['roofline']['frac'])"
python bench.py --no-cpu-baseline --no-exact-region --steps 20 --warmup 5 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver args default', round(d['value'],1), d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-exact-region --steps 20 --warmup 5 --opt bound_side=0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver args bound_side=0', round(d['value'],1), d['ms_per_step'])"
done
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
