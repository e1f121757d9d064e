#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory into profiles/<tag>_*.{csv,json}.

    python tools/summarize_pmc.py gpurun_out/prof_r1 r1
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("ggl::", "")
    return name.strip()


stats = os.path.join(src, "stats", "bench_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))

out = {}
for sub in ("fetch", "write", "sq", "tcc"):
    path = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kname, ctrs in agg.items():
        for c, vals in ctrs.items():
            out.setdefault(kname, {})[c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
# HBM traffic per launch in bytes.  FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (section HBM): on gfx950
# FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced stream of 16 B per lane -- `global_load_dwordx4` and
# `global_load_lds_dwordx4` alike.  k_symm_dl moves its operands with 16-byte-per-lane DMA, so its FETCH_SIZE is doubled
# (calibration on k_symm_dl itself: ggl_dev_symm_bench at K=32, p=500 with A == B reads 64.0 MB of unique operand bytes
# and reports FETCH_SIZE = 31.5e3 KiB).  The 8-byte-per-lane kernels (k_symm_tn, the elementwise and Theta kernels) were
# calibrated the same way at a factor of 1: k_symm_tn with A == B reports 72.2e3 KiB for the same 64.0 MB, WRITE_SIZE
# 64.7e3 KiB for a 64.0 MB output.
# Round 4: the same holds for EVERY kernel that loads 16 bytes per lane -- the per-element Theta kernels since
# round 2 (k_theta_ggl_flat4<KQ> and flat4v<8,...>; the 16 x 16-wave instance flat4v<16,...> for K > 128 uses 8-byte accesses),
# the instance copies, the persistent chain and the int8 product kernel.  (r3_pmc_summary.json reported 224 MB per launch for
# flat4v against 320 MB algorithmic: FETCH 188 MB was half of the truth; 2 x 94 + 125 = 313 MB.)
FETCH_FACTOR = {"k_symm_dl": 2.0, "k_theta_ggl_flat4v<16": 1.0, "k_theta_ggl_flat4": 2.0, "k_copy_instances": 2.0,
                "k_omega_chain": 2.0, "k_symm_i8": 2.0}
for kname, c in out.items():
    f = c.get("FETCH_SIZE", {}).get("mean_per_launch")
    w = c.get("WRITE_SIZE", {}).get("mean_per_launch")
    if f is not None and w is not None:
        fac = next((v for k, v in FETCH_FACTOR.items() if kname.startswith(k)), 1.0)
        c["fetch_correction_factor"] = fac
        c["hbm_bytes_per_launch"] = (fac * f + w) * 1024.0
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items()}, indent=1))
