"""FP64 VALU || FP64 MFMA co-issue probe: does v_fma_f64 retire beside v_mfma_f64_16x16x4_f64?
    python -m gglasso_amd.build --dev && python tools/probe_coissue.py   (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib

lib = _lib.load_dev()
out = np.zeros(12)
for rep in range(2):
    _lib.check(lib.ggl_dev_coissue_probe(_lib.ptr(out)))
print(f"MFMA only (2 waves/SIMD)            : {out[0]:6.1f} TF/s")
print(f"v_fma_f64 only (2 waves/SIMD)       : {out[1]:6.1f} TF/s")
print(f"split waves (1 MFMA + 1 DFMA / SIMD): MFMA {out[2]:6.1f} + DFMA {out[3]:6.1f} = {out[2] + out[3]:6.1f} TF/s")
for i, nv in enumerate((4, 8, 16, 32)):
    a, b = out[4 + 2 * i], out[5 + 2 * i]
    print(f"same wave, {nv:2d} v_fma_f64 per MFMA    : MFMA {a:6.1f} + DFMA {b:6.1f} = {a + b:6.1f} TF/s")
