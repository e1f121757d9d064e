#!/usr/bin/env python3
"""Per-workgroup timeline of the 64x64 symmetric-product kernel (dev tool)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
lib = _lib.load_dev()      # libggl_hip_dev.so: python -m gglasso_amd.build --dev
K, p = int(sys.argv[1]), int(sys.argv[2])
mx = 20000
buf = (ctypes.c_longlong * (mx * 5))()
nb = ctypes.c_int(0)
_lib.check(lib.ggl_dev_symm_timeline(K, p, buf, mx, ctypes.byref(nb)))
t = np.frombuffer(buf, dtype=np.int64)[: nb.value * 5].reshape(-1, 5)
t = t[t[:, 3] > 0]
xcc = t[:, 4].astype(int)
print(f"K={K} p={p}: {len(t)} workgroups")
print("k-loop cycles  (5/50/95%):", np.percentile(t[:, 2] - t[:, 1], [5, 50, 95]).astype(int))
print("epilogue cycles(5/50/95%):", np.percentile(t[:, 3] - t[:, 2], [5, 50, 95]).astype(int))
for x in range(8):
    m = xcc == x
    if not m.any():
        continue
    s0 = t[m, 0].min()
    start, end = t[m, 0] - s0, t[m, 3] - s0
    span = end.max()
    first = start < 20000
    grid = np.linspace(0, span, 11)
    conc = [int(np.sum((start <= g) & (end > g))) for g in grid]
    print(f"XCC {x}: {m.sum()} WGs, span {span} cyc; started within 20k cyc: {int(first.sum())}; "
          f"late starters begin at median {int(np.median(start[~first])) if (~first).any() else 0}; resident over time {conc}")
