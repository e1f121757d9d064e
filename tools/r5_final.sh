#!/bin/bash
# last GPU call of the round: the whole GPU suite and the bench line on the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m gglasso_amd.build --dev > /dev/null
( time python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu_final.txt 2>&1
tail -14 $O/pytest_gpu_final.txt
python bench.py > $O/bench_final_tree.log 2>&1
grep "^{" $O/bench_final_tree.log > $O/bench_final_tree.json
head -c 600 $O/bench_final_tree.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_final_tree_driver_args.json
python -c "
import json
for f in ('bench_final_tree','bench_final_tree_driver_args'):
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_step'))"
python -c "import __graft_entry__ as g; g.smoke()"
