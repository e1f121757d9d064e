#!/usr/bin/env python3
"""Persistent-chain probe (dev tool): the Omega-chain's 8 dependent symmetric products of a K-batch as
8 launches against ONE cooperative launch with grid-wide barriers between the products (csrc/gemm_sym.hip,
k_symm_chain_probe: the product kernel's own tile body in a persistent loop).  Needs libggl_hip_dev.so."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()
NPROD = 8
print(f"{NPROD} dependent products X <- X X; us per chain (per product)")
for (K, p, variants) in ((2, 500, (20,)), (4, 500, (20, 16)), (8, 500, (20, 16)), (16, 500, (20, 16, 17)), (32, 500, (17,)),
                         (20, 200, (20,)), (4, 1000, (20, 16))):
    for v in variants:
        for two_level in (0, 1):
            out = np.zeros(8)
            _lib.check(lib.ggl_dev_chain_probe(K, p, v, NPROD, 20, two_level, ptr(out)))
            a, b = out[0] * 1e3, out[1] * 1e3
            print(f"K={K:3d} p={p:4d} variant {v:2d} barrier {'per-XCD write-back' if two_level else 'per-workgroup fence'}: "
                  f"launches {a:8.1f} ({a / NPROD:6.1f})   persistent {b:8.1f} ({b / NPROD:6.1f})   ratio {b / max(a, 1e-9):5.2f}   "
                  f"grid {int(out[2]):5d}   max|diff| {out[3]:.1e}   flag {int(out[4])}", flush=True)
