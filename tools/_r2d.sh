cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
python -m pytest tests/test_gpu_ops.py tests/test_gpu_admm.py tests/test_gpu_dispatch.py -m gpu -q -x -k "bound or dispatch or speculative or sharded or fixed_length or converged or mid_sizes or extreme" > gpurun_out/r2d/pytest.txt 2>&1
tail -5 gpurun_out/r2d/pytest.txt
bash tools/ab_bench.sh r2d 3 "--opt pipeline=0 --opt fused_bounds=0 --opt fused_start=0" "--opt pipeline=0 --opt fused_bounds=0" "--opt pipeline=0" ""
