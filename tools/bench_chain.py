"""k_omega_chain against the same chain as launches (GGL_DEV build).   python tools/bench_chain.py [K p nprod]..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib

lib = _lib.load_dev()
cases = [(32, 500, 7), (16, 500, 7), (8, 500, 7), (64, 500, 7), (32, 1000, 7), (40, 200, 7)]
if len(sys.argv) > 1:
    a = [int(v) for v in sys.argv[1:]]
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for K, p, nprod in cases:
    out = np.zeros(5 + K)
    _lib.check(lib.ggl_dev_chain_run(K, p, nprod, 20, _lib.ptr(out)))
    T = (p + 63) // 64
    total = nprod * T * (T + 1) // 2
    bad = [int(k) for k in range(K) if out[5 + k] != total]
    print(f"K={K:3d} p={p:4d} {nprod} products: launches {out[0]*1e3/nprod:7.1f} us/product   chain {out[1]*1e3/nprod:7.1f} us/product "
          f"({int(out[2])} persistent workgroups)  max|diff| {out[3]:.1e}  incomplete flag {int(out[4])}  unfinished instances {bad}",
          flush=True)
