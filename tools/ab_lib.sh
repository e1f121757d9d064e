#!/bin/bash
# Runs on the GPU box: interleaved A/B of two BUILDS of libggl_hip.so inside ONE box (box-to-box noise is ~2 %).
#   tools/ab_lib.sh <tag> <rounds> <libA.so> <libB.so> ["<bench args>" ...]
# The libraries are copied over gglasso_amd/lib/libggl_hip.so in turn; the library that was installed before the run is
# restored on exit (whatever the exit path), so later pytest / bench runs in the same tree measure what build.py built.
set -u
TAG=$1; ROUNDS=$2; LA=$3; LB=$4; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
ORIG=$(mktemp /tmp/libggl_hip.orig.XXXXXX)
cp gglasso_amd/lib/libggl_hip.so $ORIG
trap 'cp $ORIG gglasso_amd/lib/libggl_hip.so; rm -f $ORIG' EXIT
[ $# -eq 0 ] && set -- ""
for r in $(seq 1 $ROUNDS); do
  for side in A B; do
    if [ $side = A ]; then cp $LA gglasso_amd/lib/libggl_hip.so; else cp $LB gglasso_amd/lib/libggl_hip.so; fi
    i=0
    for cfg in "$@"; do
      python bench.py --no-cpu-baseline --steps 60 --warmup 10 $cfg 2>/dev/null | grep "^{" >> $O/$side$i.jsonl
      i=$((i+1))
    done
  done
done
python - "$O" "$@" <<'PY'
import json, sys, statistics
o = sys.argv[1]; cfgs = sys.argv[2:]
for i, c in enumerate(cfgs):
    for side in "AB":
        rows = [json.loads(l) for l in open(f"{o}/{side}{i}.jsonl")]
        v = [r["value"] for r in rows]
        ph = rows[-1].get("phases_ms_per_step", {})
        print(f"{side} [{c}]: median {statistics.median(v):8.1f} it/s  min {min(v):8.1f} max {max(v):8.1f}  n={len(v)}  eig_omega {ph.get('eig_omega')}")
PY
