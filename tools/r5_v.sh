#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 600 python tools/event_timeline.py ggl_K32_p500 1 2>&1 | tail -40
timeout 600 python tools/event_timeline.py ggl_K32_p500 1 join_flag=0 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_admm.py -x -q -k "pipelined or speculative or side_stream or sharded" > $O/pytest_v.txt 2>&1
tail -3 $O/pytest_v.txt
{
echo "bench.py [--workload w] --opt join_flag=1|0 (GGL_OPT_JOIN_FLAG), interleaved in one box: it/s"
for rep in 1 2 3 4; do for w in ggl_K32_p500 ggl_K8_p500; do for jf in 1 0; do
  timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --regions 7 --no-cpu-baseline --no-exact-region --opt join_flag=$jf 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w join_flag=$jf', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')"
done; done; done
for rep in 1 2 3; do for jf in 1 0; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-region --opt join_flag=$jf 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver args join_flag=$jf', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')"
done; done
} > $O/join_flag_ab.txt 2>&1
cat $O/join_flag_ab.txt
