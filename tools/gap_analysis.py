#!/usr/bin/env python3
"""Per-iteration GPU timeline from a rocprofv3 --kernel-trace csv (dev tool): busy time, idle gaps and which
kernel each gap follows.   tools/gap_analysis.py <kernel_trace.csv> [iters_to_skip]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# an iteration ends with its norm reduction: the next launch starts the following one (k_form_W is no marker any more: with
# GGL_OPT_FUSED_W the Theta kernel writes W and steady-state iterations have no such launch)
# ... and with GGL_OPT_REDUCE_RIDER the reduction rides in the next A' launch: the marker is the Theta kernel, plus the
# reduction where one follows it)
starts = []
for i, r in enumerate(rows):
    if "k_theta" in r[2]:
        j = i + 2 if (i + 1 < len(rows) and "reduce_partials" in rows[i + 1][2]) else i + 1
        if j < len(rows):
            starts.append(j)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
its = list(zip(starts[skip:-1], starts[skip + 1:]))
tot = busy = 0.0
gap_after = defaultdict(float)
gap_cnt = defaultdict(int)
ktime = defaultdict(float)
kcnt = defaultdict(int)
for a, b in its:
    seg = rows[a:b + 1]
    tot += seg[-1][0] - seg[0][0]
    end = seg[0][0]
    prev = None
    for s, e, n in seg[:-1]:
        if prev is not None and s > end:
            gap_after[prev] += s - end
            gap_cnt[prev] += 1
        if e > end:
            busy += e - max(s, end)
            end = e
            prev = n
        ktime[n] += e - s
        kcnt[n] += 1
    s = seg[-1][0]
    if s > end:
        gap_after[prev] += s - end
        gap_cnt[prev] += 1
n = len(its)
print(f"{n} iterations: {tot/n/1e3:.1f} us per iteration, GPU busy {busy/n/1e3:.1f} us ({100*busy/tot:.1f} %)")
print("idle gaps by preceding kernel (us per iteration, count per iteration):")
for k, v in sorted(gap_after.items(), key=lambda kv: -kv[1]):
    print(f"  {v/n/1e3:8.1f}  {gap_cnt[k]/n:5.1f}  {k}")
print("kernel time (us per iteration, launches per iteration, us per launch):")
for k, v in sorted(ktime.items(), key=lambda kv: -kv[1]):
    print(f"  {v/n/1e3:8.1f}  {kcnt[k]/n:5.1f}  {v/kcnt[k]/1e3:8.1f}  {k}")
