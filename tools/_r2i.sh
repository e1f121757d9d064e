cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $O
( time python -m pytest tests -m gpu -q -x --durations=8 ) > $O/pytest.txt 2>&1
tail -14 $O/pytest.txt
python tools/bench_small_batches.py > $O/small_batches.txt 2>&1
cat $O/small_batches.txt
bash tools/ab_bench.sh r2i 3 "--opt pipeline=0 --opt fused_bounds=0 --opt fused_start=0 --opt theta_flat=1" ""
