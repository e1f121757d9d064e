#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
python -m gglasso_amd.build --dev > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
( timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_ops.py -q -p no:cacheprovider -k "groups or rank_two_tier or rank_deflation" ) > $O/pytest_fix.txt 2>&1; tail -15 $O/pytest_fix.txt
bash tools/r6_e.sh
