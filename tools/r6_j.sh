#!/bin/bash
# Runs on the GPU box (via gpurun): last check of the final tree -- GPU suite, smoke(), default bench line, driver-args line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6j; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py 2>&1 | grep "^{" > $O/bench_final.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_driver_args.json
python - <<'PY'
import json
for f in ("bench_final", "bench_driver_args"):
    d = json.load(open(f"gpurun_out/r6j/{f}.json"))
    print(f, round(d["value"], 1), d["unit"], d["roofline"]["frac"], d.get("cpu_baseline"))
PY
