cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_exit_checks.py tests/test_gpu_ext.py tests/test_gpu_latent_rank.py tests/test_gpu_omega_lds.py tests/test_gpu_ops.py tests/test_gpu_selection.py -q -m gpu -W always 2>&1 | grep -E "passed|failed|solver error|AssertionError|array\(" | head -12
done
