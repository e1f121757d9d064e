#!/usr/bin/env python3
"""Times every tile variant of the symmetric-product kernel (dev tool)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()      # libggl_hip_dev.so: python -m gglasso_amd.build --dev
for (K, p) in ((32, 500), (4, 500), (20, 200), (1, 1000), (32, 1000)):
    for v in range(6):
        ms = np.zeros(1)
        _lib.check(lib.ggl_dev_symm_bench(K, p, v, 20, ptr(ms)))
        tf = K * p ** 3 / (ms[0] * 1e-3) / 1e12
        print(f"K={K:3d} p={p:4d} variant {v}: {ms[0]:8.4f} ms  {tf:6.2f} TF/s algorithmic (K p^3 / t)", flush=True)
