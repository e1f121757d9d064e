"""Isolated launches of 64x64 direct-to-LDS product-kernel instances (GGL_DEV build), us per launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()
def t(K, p, v):
    ms = np.zeros(1); _lib.check(lib.ggl_dev_symm_bench(K, p, v, 40, ptr(ms))); return ms[0] * 1e3
names = {16: "<16,2> 32K", 17: "<16,3> 48K", 18: "<16,4> 64K", 19: "<32,2> 64K", 38: "<8,4> 32K", 39: "<8,6> 48K"}
print(f"{'':14s}" + "".join(f"{n:>13s}" for n in names.values()) + "   (us per launch)")
for (K, p) in ((32, 500), (16, 500), (64, 500), (128, 500), (32, 1000)):
    print(f"K={K:3d} p={p:4d}: " + "".join(f"{t(K, p, v):13.1f}" for v in names), flush=True)
