cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $O
python -m pytest tests/test_gpu_admm.py tests/test_gpu_ops.py -m gpu -q -x -k "sharded or prox_p or g6 or asymmetric or speculative" 2>&1 | tail -3
for w in ggl_K4_p500 ggl_K8_p500; do
  GGL_BENCH_FORCE_DIST=1 python bench.py --workload $w --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --comm capi 2>&1 | grep "^{" > $O/workload_${w}_sharded_1rank_rccl_capi.json
  python -c "
import json; d=json.load(open('$O/workload_${w}_sharded_1rank_rccl_capi.json')); print('$w sharded capi', round(d['value'],1), d['phases_ms_per_step'])"
done
