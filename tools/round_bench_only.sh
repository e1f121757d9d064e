#!/bin/bash
# Re-runs only the bench lines of tools/round_artifacts.sh (no pytest, no rocprofv3).   tools/round_bench_only.sh <tag>
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench_final.log 2>&1
grep "^{" $O/bench_final.log > $O/bench_final.json
for w in ggl_K20_p200 ggl_K4_p500 ggl_K32_p1000 fgl_K50_p500_latent ggl_K256_p1000; do
  python bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_$w.json
done
GGL_NS_MODE=2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_ggl_K32_p500_stable.json
python bench.py --eig 2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_ggl_K32_p500_rocsolver.json
head -c 300 $O/bench_final.json
