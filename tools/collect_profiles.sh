#!/bin/bash
# After `gpurun -- bash tools/round_artifacts.sh <tag>`: copy what the docs cite from gpurun_out/ (scratch) into profiles/ (tracked).
#   tools/collect_profiles.sh <tag>
# Every profiles/<tag>_X that has a newer gpurun_out/<tag>/X is refreshed; the rocprofv3 summaries come from tools/summarize_pmc.py.
TAG=${1:-r5}
cd "$(dirname "$0")/.."
for f in profiles/${TAG}_*; do
  x=${f#profiles/${TAG}_}
  s=gpurun_out/$TAG/$x
  [ -f "$s" ] && [ "$s" -nt "$f" ] && [ -s "$s" ] && ! grep -q "Traceback" "$s" && { cp "$s" "$f"; echo "refreshed $f"; }
done
[ -f gpurun_out/$TAG/bench_driver_args.json ] && cp gpurun_out/$TAG/bench_driver_args.json profiles/${TAG}_bench_driver_args_steps20.json
[ -f gpurun_out/$TAG/pytest_gpu.txt ] && tail -16 gpurun_out/$TAG/pytest_gpu.txt > profiles/${TAG}_pytest_gpu_tail.txt
python tools/summarize_pmc.py gpurun_out/prof_$TAG $TAG
