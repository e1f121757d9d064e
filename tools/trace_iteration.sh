#!/bin/bash
# kernel-by-kernel timeline of steady-state ADMM iterations of a bench.py workload (rocprofv3 kernel trace):
#   tools/trace_iteration.sh <workload> [extra bench.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-ggl_K4_p500}; shift
O=$R/gpurun_out/trace_$W${TRACE_TAG:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --regions 2 --no-cpu-baseline --no-exact-region "$@" > $O/run.log 2>&1
cd $R
python3 - <<PY
import csv, glob, re
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("ggl::", "")
    return n[:58]
# two iterations shortly before the end of the second timed region; an iteration ends with its norm reduction (k_form_W is
# no marker any more: with GGL_OPT_FUSED_W the Theta kernel writes W and steady-state iterations have no such launch)
# ... and with GGL_OPT_REDUCE_RIDER the reduction rides in the next A' launch: the Theta kernel, plus a reduction that follows it)
idx = []
for i, r in enumerate(rows):
    if "k_theta" in r["Kernel_Name"]:
        idx.append(i + 2 if (i + 1 < len(rows) and "k_reduce_partials" in rows[i + 1]["Kernel_Name"]) else i + 1)
a, b = idx[-16], idx[-14]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print(f"two consecutive iterations of $W${TRACE_TAG:-} (us from the first kernel's start; gap = idle time of the device before the kernel)")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f}  {short(r['Kernel_Name']):58s} {(e - s) / 1e3:7.1f} us   gap {(s - prev_end) / 1e3:6.1f}   queue {r.get('Queue_Id')}")
    prev_end = max(prev_end, e)
print(f"iteration period: {(int(rows[b]['Start_Timestamp']) - t0) / 2e3:.1f} us")
PY
