#!/usr/bin/env python3
"""Launches the symmetric-product kernel a few times (target for rocprofv3 --pmc runs)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
K, p, v, iters = (int(x) for x in sys.argv[1:5])
lib = _lib.load_dev()      # libggl_hip_dev.so: python -m gglasso_amd.build --dev
ms = np.zeros(1)
_lib.check(lib.ggl_dev_symm_bench(K, p, v, iters, ptr(ms)))
print(f"K={K} p={p} variant {v}: {ms[0]:.4f} ms")
