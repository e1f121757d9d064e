#!/bin/bash
# Runs on the GPU box: interleaved A/B of bench.py configurations inside ONE box (box-to-box noise is ~2 %).
#   tools/ab_bench.sh <tag> <rounds> "<args A>" "<args B>" ...
set -u
TAG=$1; ROUNDS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for r in $(seq 1 $ROUNDS); do
  i=0
  for cfg in "$@"; do
    python bench.py --no-cpu-baseline --steps 60 --warmup 10 $cfg 2>/dev/null | grep "^{" >> $O/cfg$i.jsonl
    i=$((i+1))
  done
done
python - "$O" "$@" <<'PY'
import json, sys, glob, statistics
o = sys.argv[1]; cfgs = sys.argv[2:]
for i, c in enumerate(cfgs):
    rows = [json.loads(l) for l in open(f"{o}/cfg{i}.jsonl")]
    v = [r["value"] for r in rows]
    ph = rows[-1].get("phases_ms_per_step", {})
    print(f"cfg{i} [{c}]: median {statistics.median(v):8.1f} it/s  min {min(v):8.1f} max {max(v):8.1f}  n={len(v)}  phases {ph}")
PY
