#!/bin/bash
# Runs on the GPU box (via gpurun): the round-6 judged artifacts in one call -> gpurun_out/r6/, gpurun_out/prof_r6/
set -u
TAG=r6
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -m gglasso_amd.build --dev > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
( time timeout 2400 python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1; cd $R
python bench.py > $O/bench_final.log 2>&1; grep "^{" $O/bench_final.log > $O/bench_final.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_driver_args.json
for w in ggl_K20_p200 ggl_K4_p500 ggl_K8_p500 ggl_K16_p500 ggl_K32_p1000 fgl_K50_p500_latent ggl_K256_p1000 ggl_K64_p100 ggl_K256_p64 ggl_K32_p128 ggl_K32_p501 ggl_K20_p201; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_$w.json
done
timeout 600 python bench.py --workload sgl_p1000_grid20 > $O/c2.log 2>&1; grep "^{" $O/c2.log > $O/workload_sgl_p1000_grid20.json
python bench.py --opt ns_tol=0 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_ggl_K32_p500_exact_omega_step.json
for w in ggl_K4_p500 ggl_K8_p500 ggl_K16_p500; do
  GGL_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --comm capi 2>&1 | grep "^{" > $O/workload_${w}_sharded_1rank_rccl_capi.json
done
python tools/bench_grid.py 2>&1 | grep "^{" > $O/workload_sgl_grid_p1000_L20.json
python tools/bench_mgl_grid.py 2>&1 | grep "^{" > $O/workload_mgl_grid_8x1_K4_p500.json
{ for a in "--p 50 --points 20" "--p 64 --points 100"; do python tools/bench_grid.py $a --no-sequential 2>&1 | grep "^{"; done; } > $O/grids.txt 2>&1
timeout 600 python tools/stress_rank_table.py 20 500 127 > $O/stress_rank_table_poison127.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/c2 -o bench -- python3 $R/bench.py --workload sgl_p1000_grid20 --steps 20 --warmup 5 --regions 2 --no-cpu-baseline > $O/c2_prof.log 2>&1 )
python - <<PY > $O/c2_kernel_stats.txt 2>&1
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$TAG/c2/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
rm -rf $R/gpurun_out/prof_$TAG/*/*.db 2>/dev/null
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
for f in $O/bench_final.json $O/bench_driver_args.json $O/workload_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d.get('value', 0), 1), d.get('unit'), (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
