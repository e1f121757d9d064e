#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
python -m gglasso_amd.build --dev > $O/build.log 2>&1
( time GGL_DEBUG_POISON=71 timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_poison71.txt 2>&1; tail -6 $O/pytest_poison71.txt
python bench.py > $O/bench_final.log 2>&1; grep "^{" $O/bench_final.log > $O/bench_final.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_driver_args.json
for f in bench_final bench_driver_args; do python -c "import json;d=json.load(open('$O/$f.json'));print('$f',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('traffic_source'))"; done
for w in ggl_K64_p100 ggl_K20_p200; do python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_$w.json; python -c "import json;d=json.load(open('$O/workload_$w.json'));print('$w',d['value'])"; done
