#!/usr/bin/env python3
"""Where the direct-to-LDS product kernel's time goes (dev tool, libggl_hip_dev.so): isolated launches of the three-stage
64x64 kernel (variant 17) and the 32x32 kernel (variant 20) against their timing ablations -- no operand loads, no MFMA,
no epilogue, no slab barrier (wrong results, same instruction stream otherwise)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()
def t(K, p, v):
    ms = np.zeros(1); _lib.check(lib.ggl_dev_symm_bench(K, p, v, 40, ptr(ms))); return ms[0] * 1e3
names = ("whole", "no loads", "no MFMA", "no epilogue", "no barrier")
print(f"{'':28s}" + "".join(f"{n:>13s}" for n in names) + "   (us per launch)")
for (K, p) in ((32, 500), (16, 500), (8, 500), (4, 500), (20, 200), (8, 1000)):
    for base, abl in ((17, (30, 31, 32, 33)), (20, (34, 35, 36, 37))):
        row = [t(K, p, base)] + [t(K, p, v) for v in abl]
        print(f"K={K:3d} p={p:4d} variant {base:2d}:     " + "".join(f"{x:13.1f}" for x in row), flush=True)
