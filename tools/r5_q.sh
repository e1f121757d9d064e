cd $GRAFT_REPO_ROOT
for v in 1 0; do python bench.py --workload ggl_K64_p100 --no-cpu-baseline --opt copy_rider=$v 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print($v, round(d['value'],1), {k:d[k] for k in d if k not in ('config','roofline','metric','unit')})"; done
timeout 300 python tools/event_timeline.py ggl_K64_p100 1 2>&1 | tail -40 | cut -c1-120
