#!/bin/bash
# Runs on the GPU box (via gpurun): the round's judged artifacts in one call.
#   tools/round_artifacts.sh <tag>
# -> gpurun_out/<tag>/bench_final.json, workload_*.json, pytest_gpu.txt and gpurun_out/prof_<tag>/ (rocprofv3)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1
cd $R
python bench.py > $O/bench_final.log 2>&1
tail -1 $O/bench_final.log > $O/bench_final.json
for w in ggl_K20_p200 ggl_K4_p500 ggl_K32_p1000 fgl_K50_p500_latent ggl_K256_p1000; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/workload_$w.json
done
GGL_NS_MODE=2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 > $O/workload_ggl_K32_p500_stable.json
python bench.py --eig 2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $O/workload_ggl_K32_p500_rocsolver.json
python tools/gap_analysis.py $(find $R/gpurun_out/prof_$TAG/stats -name "*kernel_trace.csv" | head -1) 4 > $O/timeline.txt 2>&1
head -c 600 $O/bench_final.json
