"""FP64 MFMA with NV integer VALU instructions per MFMA in the same wave (libggl_hip_dev.so): does non-FP64 vector work
issue in the shadow of the matrix pipe, or does it take MFMA issue slots?   python tools/probe_mfma_mix.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GGL_MFMA_MIX_VERBOSE"] = "1"
os.environ["GGL_MFMA_PROBE_VERBOSE"] = "1"
from gglasso_amd import _lib

lib = _lib.load_dev()
out = np.zeros(1)
_lib.check(lib.ggl_dev_mfma_f64_peak(_lib.ptr(out)))
print(f"MFMA-only peak: {out[0]:.1f} TF/s")
o = np.zeros(6)
for _ in range(2):
    _lib.check(lib.ggl_dev_mfma_lds_probe(_lib.ptr(o)))
for name, v in zip(("2x2 blocks/wave, 3 WG/CU (4 reads : 4 MFMA)", "2x2, 5 WG/CU", "2x4, 3 WG/CU (6 : 8)", "4x4, 2 WG/CU (8 : 16)",
                    "4x4, 3 WG/CU", "1x1, 5 WG/CU (2 : 1)"), o):
    print(f"MFMA fed from LDS, {name:46s}: {v:6.1f} TF/s")
