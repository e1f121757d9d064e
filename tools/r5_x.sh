#!/bin/bash
# final-tree suite + bench, then the small-size event timelines (after the k_cw_final tail and the calm band of 8)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
bash tools/r5_final.sh
for w in ggl_K4_p500 ggl_K20_p200; do
  timeout 300 python tools/event_timeline.py $w 1 > $O/event_timeline_$w.txt 2>&1
  tail -3 $O/event_timeline_$w.txt
  grep -E "cw_final|bound_rows" $O/event_timeline_$w.txt | head -4
done
for w in ggl_K4_p500 ggl_K8_p500 ggl_K20_p200 ggl_K16_p500; do
  python bench.py --workload $w --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value'],1), d['ms_per_step'])"
done
