cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2h
mkdir -p $O
python -m pytest tests/test_gpu_admm.py tests/test_gpu_dispatch.py -m gpu -q -x -k "bound or speculative or sharded or fixed_length or asymmetric or dispatch" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
bash tools/ab_bench.sh r2h 3 "--opt pipeline=0 --opt fused_bounds=0 --opt fused_start=0 --opt interleave=0" "--opt interleave=0" "" "--opt theta_flat=2"
for w in ggl_K4_p500 ggl_K20_p200 ggl_K8_p500 ggl_K16_p500 ggl_K32_p1000 fgl_K50_p500_latent; do
  python bench.py --workload $w --steps 30 --warmup 8 --regions 3 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/$w.json
  python -c "
import json; d=json.load(open('$O/$w.json')); print('$w', round(d['value'],1), 'it/s', d['phases_ms_per_step'], d['roofline']['frac'], d['roofline']['frac_of_step'])"
done
