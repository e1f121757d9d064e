#!/bin/bash
# round 6, third GPU call: fixed tests, odd-p Theta kernel, dev-only options
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
python -m gglasso_amd.build --dev > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
( timeout 1500 python -m pytest tests/test_gpu_groups.py tests/test_gpu_dispatch.py tests/test_gpu_chain.py -q -p no:cacheprovider -k "groups or odd_p or c5_slab or chain" ) > $O/pytest_new.txt 2>&1; tail -30 $O/pytest_new.txt
( timeout 1500 python -m pytest tests/test_gpu_admm.py -q -x -p no:cacheprovider ) > $O/pytest_admm.txt 2>&1; tail -6 $O/pytest_admm.txt
for rep in 1 2; do for w in ggl_K32_p501 ggl_K32_p502 ggl_K20_p201 ggl_K20_p202; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/odd_${w}_$rep.json
  python -c "import json;d=json.load(open('$O/odd_${w}_$rep.json'));print('$w',round(d['value'],1),d['unit'],round(d['ms_per_step'],4),'ms', d['phases_ms_per_step'])"
done; done
