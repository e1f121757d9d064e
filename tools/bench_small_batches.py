#!/usr/bin/env python3
"""Product-kernel variants at the per-GPU batch sizes of the headline under K-sharding (dev tool):
K = 16 / 8 / 4 at p = 500 (2 / 4 / 8 GPUs), variants 9 (32x32 tiles), 16 / 17 (64x64 direct-to-LDS, 2 / 3 stages)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()      # libggl_hip_dev.so: python -m gglasso_amd.build --dev
VARIANTS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (9, 20, 27, 28, 29, 16, 17)
for (K, p) in ((16, 500), (8, 500), (4, 500), (2, 500), (20, 200), (4, 1000), (8, 1000)):
    for v in VARIANTS:
        ms = np.zeros(1)
        _lib.check(lib.ggl_dev_symm_bench(K, p, v, 30, ptr(ms)))
        print(f"K={K:3d} p={p:4d} variant {v:2d}: {ms[0]*1e3:8.1f} us  {K * p ** 3 / (ms[0] * 1e-3) / 1e12:6.2f} TF/s", flush=True)
