#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_admm.py -x -q -k "side_stream or speculative or pipelined" > $O/pytest_o.txt 2>&1
tail -3 $O/pytest_o.txt
{
echo "bench.py --workload w --opt bound_side=1|0 (GGL_OPT_BOUND_SIDE), interleaved in one box, three pairs: it/s"
for rep in 1 2 3; do for w in ggl_K4_p500 ggl_K8_p500 ggl_K20_p200 ggl_K16_p500 ggl_K32_p500 ggl_K64_p100 ggl_K32_p128; do for bs in 1 0; do
  python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --no-exact-region --opt bound_side=$bs 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w bound_side=$bs', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms')"
done; done; done
} > $O/bound_side.txt 2>&1
cat $O/bound_side.txt
