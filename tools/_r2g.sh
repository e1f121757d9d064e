cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_admm.py -m gpu -q -x -k "bound or speculative or sharded or fixed_length or asymmetric" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
bash tools/ab_bench.sh r2g 3 "--opt pipeline=0 --opt fused_bounds=0 --opt fused_start=0" "--opt pipeline=0 --opt fused_start=0" "--opt fused_bounds=0 --opt fused_start=0" "" "--opt theta_flat=2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --regions 2 --no-cpu-baseline > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/gap_analysis.py $T 8 > $O/timeline.txt 2>&1
python tools/iter_trace.py $T 30 > $O/iter.txt 2>&1
cat $O/timeline.txt $O/iter.txt
rm -rf $O/trace
bash tools/_r2f.sh
