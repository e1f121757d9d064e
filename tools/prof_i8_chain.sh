#!/bin/bash
# per-kernel durations of one int8 Omega-step chain (K=32, p=500): rocprofv3 kernel trace of tools/bench_omega_i8.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_i8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && export I8_ONE_CFG=1
rocprofv3 --kernel-trace --output-format csv -d $O -o i8 -- python3 $R/tools/bench_omega_i8.py 500 ${1:-32} > $O/run.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_i8/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "i8" in r["Kernel_Name"]]
for r in rows[-8:]:
    print(f'{r["Kernel_Name"][:80]:80s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  vgpr {r.get("VGPR_Count")} lds {r.get("LDS_Block_Size")} scratch {r.get("Scratch_Size")}')
PY
