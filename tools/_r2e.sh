cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_ops.py tests/test_gpu_admm.py -m gpu -q -x -k "bound or speculative or sharded or fixed_length" > gpurun_out/r2e/pytest.txt 2>&1
tail -3 gpurun_out/r2e/pytest.txt
bash tools/ab_bench.sh r2e 3 "--opt pipeline=0 --opt fused_bounds=0 --opt fused_start=0" "--opt pipeline=0 --opt fused_start=0" "--opt fused_bounds=0 --opt fused_start=0" ""
O=$GRAFT_REPO_ROOT/gpurun_out/r2e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --regions 2 --no-cpu-baseline > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/gap_analysis.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 8 > $O/timeline.txt 2>&1
cat $O/timeline.txt
rm -rf $O/trace
