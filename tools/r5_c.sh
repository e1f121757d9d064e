#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_batch_isolation.py tests/test_gpu_omega_lds.py -x -q > $O/pytest_c.txt 2>&1
tail -3 $O/pytest_c.txt
python tools/time_batch.py --p 1000 --points 20 --compact 1
python tools/time_batch.py --p 1000 --points 20 --compact 0
python tools/time_batch.py --p 64 --points 100
python tools/time_batch.py --p 50 --points 20
python tools/bench_omega_lds.py 256:64 256:48 64:64 > $O/omega_lds_kernel.txt 2>&1
grep "K=" $O/omega_lds_kernel.txt | head -40
for w in ggl_K256_p64 ggl_K64_p100; do
  python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/workload_$w.json
  python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --opt omega_lds=4 2>&1 | grep "^{" > $O/workload_${w}_lds4waves.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/workload_ggl_K256_p64*.json")+glob.glob("$O/workload_ggl_K64_p100*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), d['ms_per_step'], d.get('phases_ms'))
PY
