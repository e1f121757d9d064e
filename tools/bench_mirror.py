#!/usr/bin/env python3
"""Upper bound of what dropping the mirror write of the symmetric product would buy (dev tool): variant 16
(direct-to-LDS, double buffer) vs variant 21 (same without the mirror pass; wrong lower triangle)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()      # libggl_hip_dev.so: python -m gglasso_amd.build --dev
for (K, p) in ((32, 500), (16, 500), (32, 1000)):
    for v in (16, 21, 17):
        ms = np.zeros(1)
        _lib.check(lib.ggl_dev_symm_bench(K, p, v, 30, ptr(ms)))
        print(f"K={K:3d} p={p:4d} variant {v}: {ms[0]*1e3:8.1f} us  {K * p ** 3 / (ms[0] * 1e-3) / 1e12:6.2f} TF/s", flush=True)
