#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_latent_rank.py -x -q -k "rank or latent or deflat" > $O/pytest_k.txt 2>&1
tail -3 $O/pytest_k.txt
for cw in 0 1; do for l0 in 2e-3 3e-3 4e-3 6e-3; do
  python bench.py --workload fgl_K50_p500_latent --steps 30 --warmup 8 --regions 3 --no-cpu-baseline --opt rank_cw=$cw --opt rank_l0_deflate=$l0 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); ns=d['newton_schulz']; print('C4 rank_cw=$cw l0_deflate=$l0', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms  L-step products', round(ns['lstep_products_per_call'],2), ns['lstep_two_tier_totals'], 'retries', ns['lstep_retries'], 'eigh', ns['lstep_eigh_fallbacks'], 'parity', d.get('parity'))"
done; done
python -m pytest tests/test_gpu_dispatch.py -x -q -k "c4" > $O/pytest_k2.txt 2>&1
tail -3 $O/pytest_k2.txt
