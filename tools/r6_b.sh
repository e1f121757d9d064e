#!/bin/bash
# round 6, second GPU call: odd p on the DMA kernel, group schedules, the new tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
python -m gglasso_amd.build --dev > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "symm_product_kernel or bound_partials" -p no:cacheprovider ) > $O/pytest_ops.txt 2>&1; tail -5 $O/pytest_ops.txt
( timeout 1200 python -m pytest tests/test_gpu_groups.py tests/test_gpu_dispatch.py -q -p no:cacheprovider -k "groups or odd_p or c2_ or c3_ or c5_slab" ) > $O/pytest_new.txt 2>&1; tail -40 $O/pytest_new.txt
for w in ggl_K32_p501 ggl_K32_p502 ggl_K20_p201 ggl_K20_p202 ggl_K32_p500; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/after_$w.json
  python -c "import json;d=json.load(open('$O/after_$w.json'));print('$w',round(d['value'],1),d['unit'],round(d['ms_per_step'],4),'ms', (d.get('roofline') or {}).get('frac'), d['config'].get('product_kernel'))"
done
for g in 1 0 1 0; do
  timeout 300 python bench.py --workload sgl_p1000_grid20 --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --opt group_sched=$g 2>/dev/null | grep "^{" > $O/c2_group$g.json
  python -c "import json;d=json.load(open('$O/c2_group$g.json'));print('c2 group_sched=$g',round(d['value'],1),d['unit'],round(d['ms_per_step'],4),'ms', d['newton_schulz']['group_schedules'], (d.get('value_exact') or {}).get('products_per_step'))"
done
( timeout 900 python -m pytest tests/test_gpu_selection.py tests/test_gpu_latent_rank.py tests/test_gpu_batch_isolation.py -q -p no:cacheprovider ) > $O/pytest_sel.txt 2>&1; tail -8 $O/pytest_sel.txt
