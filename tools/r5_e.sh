#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
bash tools/trace_iteration.sh ggl_K32_p500 > $O/timeline_headline.txt 2>&1
TRACE_TAG=_nofusedw bash tools/trace_iteration.sh ggl_K32_p500 --opt fused_w=0 > $O/timeline_headline_nofusedw.txt 2>&1
bash tools/trace_iteration.sh ggl_K8_p500 > $O/timeline_K8.txt 2>&1
GGL_BENCH_FORCE_DIST=1 TRACE_TAG=_sharded bash tools/trace_iteration.sh ggl_K8_p500 --comm capi > $O/timeline_K8_sharded.txt 2>&1
bash tools/trace_iteration.sh ggl_K20_p200 > $O/timeline_K20_p200.txt 2>&1
bash tools/trace_iteration.sh ggl_K4_p500 > $O/timeline_K4.txt 2>&1
tail -4 $O/timeline_*.txt
rm -rf $R/gpurun_out/trace_*/*/*.db 2>/dev/null
find $R/gpurun_out/trace_* -name "*.csv" -size +5M -delete 2>/dev/null
