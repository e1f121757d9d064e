#!/usr/bin/env python3
"""One symmetric product launch (csrc/gemm_sym.hip) at SMALL sizes, every product-kernel variant: what symm_auto_variant
should pick below p = 130.   python tools/bench_symm_small.py [K:p ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr

lib = _lib.load()
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(64, 100), (32, 128), (5, 100), (10, 100), (64, 80), (128, 96), (20, 200),
                                                                        (64, 40), (256, 64), (16, 32), (8, 16), (256, 10), (3, 64), (50, 66)]
for K, p in cases:
    row = []
    for v in (-1, 0, 9, 16, 17, 20):
        ms = np.zeros(1)
        rc = lib.ggl_dev_symm_bench(K, p, v, 50, ptr(ms))
        row.append(f"v{v}: {ms[0] * 1e3:6.1f}" if rc == 0 else f"v{v}:    n/a")
    print(f"K={K:4d} p={p:4d}  us per launch  " + "  ".join(row))
