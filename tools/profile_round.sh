#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and the two HBM-traffic PMC passes of the SAME
# bench.py command, each in its own rocprofv3 run (counters never share a run with --stats).
#   tools/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --regions 3 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o bench -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o bench -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o bench -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -o bench -- python3 $R/bench.py $ARGS > $O/tcc.log 2>&1
tail -1 $O/stats.log
ls $O/*
