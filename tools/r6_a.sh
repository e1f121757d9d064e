#!/bin/bash
# round 6, first GPU call: hunt for the intermittent latent-grid failure (poison patterns), vendor yardstick, "before" numbers
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
for pz in 127 71 1 0; do
  timeout 600 python tools/stress_rank_table.py 12 500 $pz > $O/stress_poison_$pz.txt 2>&1; echo "stress poison $pz rc $?"; tail -3 $O/stress_poison_$pz.txt
done
timeout 300 python tools/bench_vendor.py > $O/vendor_yardstick.txt 2>&1; tail -50 $O/vendor_yardstick.txt
for w in ggl_K32_p500 ggl_K32_p501 ggl_K32_p502 ggl_K20_p201 ggl_K20_p202 sgl_p1000_grid20; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/before_$w.json
  python -c "import json;d=json.load(open('$O/before_$w.json'));print('$w',round(d['value'],1),d['unit'],round(d['ms_per_step'],4),'ms', (d.get('roofline') or {}).get('frac'))"
done
( time GGL_DEBUG_POISON=127 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_poison127.txt 2>&1
tail -30 $O/pytest_poison127.txt
