#!/usr/bin/env python3
"""One iteration of a rocprofv3 --kernel-trace csv, launch by launch (dev tool): start offset, duration, queue and
name of every kernel between two k_form_W launches.   tools/iter_trace.py <kernel_trace.csv> [iteration index]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                     r["Kernel_Name"].split("(")[0][:70]))
rows.sort()
starts, armed = [], True
for i, r in enumerate(rows):
    if "form_W" in r[3] and armed:
        starts.append(i)
        armed = False
    elif "reduce_partials" in r[3]:
        armed = True
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[it], starts[it + 1]
t0 = rows[a][0]
for s, e, q, n in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{q}  {n}")
print(f"iteration: {(rows[b][0] - t0) / 1e3:.1f} us")
