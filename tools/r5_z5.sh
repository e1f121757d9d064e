#!/bin/bash
# GGL_OPT_PARTS_ORDER at the two-part workloads: interleaved, four rounds; the bitwise test first
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
GGL_TEST_OPTIONS="parts_order=1" timeout 600 python -m pytest tests/test_gpu_admm.py -x -q -m gpu -k "side_stream or pipelined" 2>&1 | grep -E "assert|Error|passed|failed" | head -6
: > $O/parts_order_ab.txt
for rep in 1 2 3 4; do
  for w in ggl_K32_p500 ggl_K8_p500 ggl_K32_p1000; do
    for b in 1 0; do
      python bench.py --workload $w --no-cpu-baseline --opt parts_order=$b 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w parts_order=$b', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" >> $O/parts_order_ab.txt
    done
  done
done
sort -s -k1,2 $O/parts_order_ab.txt
