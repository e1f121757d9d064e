#!/bin/bash
# PMC passes over isolated launches of the product kernel (tools/run_symm.py): where do its waves wait?   tools/pmc_symm.sh <tag>
set -u
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_WAVE32_LDS SQ_WAVES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o s -- python3 $R/tools/run_symm.py 32 500 17 10 > $O/p$i.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"].split("(")[0][-40:]
        acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
for nm, cs in acc.items():
    print(nm)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
