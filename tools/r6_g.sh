#!/bin/bash
# round 6: does the grouping rule cost the launch-bound sizes anything?  group_sched = 1 / 0 interleaved, two rounds
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
python -m gglasso_amd.build > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
for rep in 1 2; do for w in ggl_K64_p100 ggl_K20_p200 ggl_K32_p128 ggl_K4_p500 ggl_K256_p1000; do for g in 1 0; do
  timeout 400 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --no-exact-region --opt group_sched=$g 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w group_sched=$g', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')"
done; done; done | tee $O/group_sched_small_sizes.txt
for g in 1 2 3; do timeout 400 python bench.py --workload sgl_p1000_grid20 --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --no-exact-region --opt group_sched=$g 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 group_sched=$g', round(d['value'],1), 'it/s', d['newton_schulz']['group_schedules'])"; done | tee -a $O/group_sched_small_sizes.txt
