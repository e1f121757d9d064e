#!/bin/bash
# round 5, first GPU call: the advisor fixes' tests + the three evidence files round 4 left as tracebacks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_batch_isolation.py tests/test_gpu_latent_rank.py -x -q > $O/pytest_isolation.txt 2>&1
tail -3 $O/pytest_isolation.txt
python tools/bench_tile_variants.py > $O/tile_variants.txt 2>&1
python tools/bench_small_batches.py > $O/small_batches_product_kernel.txt 2>&1
python tools/bench_chain.py > $O/omega_chain_persistent_per_instance.txt 2>&1
tail -3 $O/tile_variants.txt $O/small_batches_product_kernel.txt $O/omega_chain_persistent_per_instance.txt
python bench.py --workload fgl_K50_p500_latent --steps 30 --warmup 8 --regions 3 --no-cpu-baseline 2>&1 | grep "^{" > $O/c4_before.json
head -c 1500 $O/c4_before.json
