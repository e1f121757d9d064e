cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
( time python -m pytest tests -m gpu -q -W always ) > gpurun_out/r5/pytest_gpu_last.txt 2>&1
tail -6 gpurun_out/r5/pytest_gpu_last.txt
grep -c "solver error" gpurun_out/r5/pytest_gpu_last.txt
