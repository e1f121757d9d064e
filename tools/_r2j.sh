cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2j
mkdir -p $O
for w in ggl_K4_p500 ggl_K8_p500; do
for o in "" "--opt parts_small=4"; do
  python bench.py --workload $w --steps 40 --warmup 8 --regions 5 --no-cpu-baseline $o 2>/dev/null | grep "^{" > $O/t.json
  python -c "
import json; d=json.load(open('$O/t.json')); print('$w [$o]', round(d['value'],1), 'it/s', d['phases_ms_per_step'], d['timed_regions'])"
done; done
python -m pytest tests/test_gpu_admm.py -m gpu -q -x -k "speculative or fused_bound" 2>&1 | tail -2
GGL_BENCH_FORCE_DIST=1 python bench.py --workload ggl_K4_p500 --steps 40 --warmup 8 --regions 5 --no-cpu-baseline --comm capi 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K4 sharded capi', round(d['value'],1), d['phases_ms_per_step'])"
GGL_BENCH_FORCE_DIST=1 python bench.py --workload ggl_K4_p500 --steps 40 --warmup 8 --regions 5 --no-cpu-baseline --comm torch 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K4 sharded torch', round(d['value'],1), d['phases_ms_per_step'])"
python tools/bench_mgl_grid.py > $O/mgl_grid.json 2>$O/mgl_grid.err; cat $O/mgl_grid.json; tail -3 $O/mgl_grid.err
