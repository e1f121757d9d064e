cd $GRAFT_REPO_ROOT
echo "--- poisoned arenas and lazy buffers: the rank-table stress"
GGL_DEBUG_POISON=1 timeout 600 python tools/stress_rank_table.py 20 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-700
echo "--- poisoned: the GPU suite (no -x)"
( GGL_DEBUG_POISON=1 timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_poison.txt 2>&1
tail -40 gpurun_out/pytest_poison.txt | cut -c1-200
