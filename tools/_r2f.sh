cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $O
python tools/bench_small_batches.py > $O/small_batches.txt 2>&1
cat $O/small_batches.txt
for w in ggl_K4_p500 ggl_K20_p200 ggl_K8_p500 ggl_K16_p500; do
  python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/$w.json
  python -c "
import json; d=json.load(open('$O/$w.json')); print('$w', round(d['value'],1), 'it/s', d['phases_ms_per_step'], d['roofline']['launches_per_step'], d['newton_schulz']['prelaunched_chains_dropped'])"
done
cd /tmp && export TMPDIR=/tmp
for w in ggl_K4_p500 ggl_K20_p200; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$w -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --regions 2 --no-cpu-baseline > $O/trace_$w.log 2>&1
  python $GRAFT_REPO_ROOT/tools/gap_analysis.py $(find $O/trace_$w -name "*kernel_trace.csv" | head -1) 8 > $O/timeline_$w.txt 2>&1
  cat $O/timeline_$w.txt
  rm -rf $O/trace_$w
done
