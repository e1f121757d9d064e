#!/bin/bash
# the three riders together: tests, then interleaved A/B default / reduce rider off / all riders off, three rounds per workload
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_admm.py -x -q -m gpu -k "side_stream or pipelined or speculative" 2>&1 | grep -E "assert|Error|passed|failed" | head -12
: > $O/riders_ab.txt
for rep in 1 2 3; do
  for w in ggl_K4_p500 ggl_K8_p500 ggl_K20_p200 ggl_K16_p500 ggl_K32_p500 ggl_K64_p100 ggl_K32_p128; do
    for v in "" "--opt reduce_rider=0" "--opt reduce_rider=0 --opt copy_rider=0 --opt cw_rider=0"; do
      python bench.py --workload $w --no-cpu-baseline $v 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w [$v]', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')" >> $O/riders_ab.txt
    done
  done
done
cat $O/riders_ab.txt | cut -c1-260
for w in ggl_K4_p500 ggl_K20_p200 ggl_K32_p500; do
  timeout 300 python tools/event_timeline.py $w 1 > $O/event_timeline_${w}_riders.txt 2>&1
  tail -4 $O/event_timeline_${w}_riders.txt | cut -c1-150
done
