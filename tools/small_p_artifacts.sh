#!/bin/bash
# The small-matrix regime (p <= 128), round 4: the LDS-resident Omega-step kernel stand-alone, the product kernels at small
# sizes, the three small-p bench workloads and the kernel statistics of the K = 256, p = 64 one.
#   tools/small_p_artifacts.sh <tag>     (on the GPU box; called by tools/round_artifacts.sh)
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/bench_omega_lds.py > $O/omega_lds_kernel.txt 2>&1
python tools/bench_symm_small.py > $O/product_kernel_small_p.txt 2>&1
{
  echo "bench.py --workload <w> [--opt omega_lds=0] --steps 30 --warmup 8 --regions 5 --no-cpu-baseline   (whole ADMM iterations per second, phases in ms per iteration)"
  for w in ggl_K256_p64 ggl_K64_p100 ggl_K32_p128 ggl_K20_p200; do
    for o in omega_lds=1 omega_lds=0; do
      python bench.py --workload $w --opt $o --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/tmp_small.json
      [ $o = omega_lds=1 ] && cp $O/tmp_small.json $O/workload_$w.json
      python - <<PY
import json
d = json.load(open("$O/tmp_small.json"))
r = d["roofline"]
print(f"$w  $o  {d['value']:8.1f} it/s  {d['phases_ms_per_step']}  {r['kernel'][:40]}  {r['frac']:.3f} of the FP64 matrix peak over the phase")
PY
    done
  done
  rm -f $O/tmp_small.json
} > $O/small_p_workloads.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/c3 -o bench -- python3 $R/bench.py --workload ggl_K256_p64 --steps 30 --warmup 8 --regions 2 --no-cpu-baseline > $O/c3_prof.log 2>&1 )
python - <<PY > $O/c3_kernel_stats.txt 2>&1
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$TAG/c3/**/*kernel_stats.csv", recursive=True)
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --workload ggl_K256_p64 --steps 30 --warmup 8 --regions 2 --no-cpu-baseline")
for r in list(csv.DictReader(open(f[0])))[:10]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
rm -rf $R/gpurun_out/prof_$TAG/c3/*.db $R/gpurun_out/prof_$TAG/c3/*/*.db 2>/dev/null
find $R/gpurun_out/prof_$TAG/c3 -name "*kernel_trace.csv" -size +20M -delete
cat $O/small_p_workloads.txt $O/c3_kernel_stats.txt
