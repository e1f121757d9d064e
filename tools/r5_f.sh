#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_admm.py tests/test_gpu_multirank.py tests/test_gpu_omega_lds.py -x -q -k "sharded or rccl or pipelined or omega or lds or kernel" > $O/pytest_f.txt 2>&1
tail -3 $O/pytest_f.txt
for w in ggl_K4_p500 ggl_K8_p500 ggl_K16_p500; do
  python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_$w.json
  GGL_BENCH_FORCE_DIST=1 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --comm capi 2>&1 | grep "^{" > $O/workload_${w}_sharded_1rank_rccl_capi.json
  GGL_BENCH_FORCE_DIST=1 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --comm capi --opt fused_w=0 2>&1 | grep "^{" > $O/workload_${w}_sharded_1rank_rccl_capi_nofusedw.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/workload_ggl_K*_p500*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), d['ms_per_step'], d.get('phases_ms'), d.get('pipeline'))
PY
python tools/bench_omega_lds.py 256:64 256:50 256:40 > $O/omega_lds_kernel.txt 2>&1
grep "K=" $O/omega_lds_kernel.txt | grep "2e-12" | head -20
python tools/bench_grid.py --p 50 --points 20 --no-sequential 2>&1 | grep "^{" | cut -c1-300
python tools/bench_grid.py --p 64 --points 100 --no-sequential 2>&1 | grep "^{" | cut -c1-300
