#!/bin/bash
# GGL_OPT_PARTS_BIAS at the headline and the K = 8 slab: interleaved, three rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
: > $O/parts_bias_ab.txt
for rep in 1 2 3; do
  for b in 0 1 2 -1 3; do
    python bench.py --workload ggl_K32_p500 --no-cpu-baseline --opt parts_bias=$b 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ggl_K32_p500 parts_bias=$b', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', 'parity', d.get('parity',{}).get('max_abs_diff'))" >> $O/parts_bias_ab.txt
  done
  for b in 0 1 -1; do
    python bench.py --workload ggl_K8_p500 --no-cpu-baseline --opt parts_bias=$b 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ggl_K8_p500 parts_bias=$b', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" >> $O/parts_bias_ab.txt
  done
done
cat $O/parts_bias_ab.txt
