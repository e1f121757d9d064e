#!/bin/bash
# round 6: the whole GPU suite on the split tree, then the headline / C2 / odd-p lines
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
python -m gglasso_amd.build --dev > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
python bench.py > $O/bench_final.log 2>&1; grep "^{" $O/bench_final.log > $O/bench_final.json; head -c 600 $O/bench_final.json; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_driver_args.json; python -c "import json;d=json.load(open('$O/bench_driver_args.json'));print('driver args',d['value'],d['ms_per_step'],d['roofline']['frac'])"
python bench.py --workload sgl_p1000_grid20 > $O/c2.log 2>&1; grep "^{" $O/c2.log > $O/workload_sgl_p1000_grid20.json; python -c "import json;d=json.load(open('$O/workload_sgl_p1000_grid20.json'));print('C2',d['value'],d['ms_per_step'],d['newton_schulz']['group_schedules'],d.get('parity'))"
