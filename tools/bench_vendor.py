#!/usr/bin/env python3
"""A measured ceiling for the symmetric-product kernel: rocBLAS FP64 GEMM / SYRK / SYRKX at the shapes the
Omega-step multiplies, next to k_symm_dl, on an EXECUTED-flop basis (dev library only; nothing on the solver's path calls rocBLAS
GEMMs).  python tools/bench_vendor.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()
names = ["rocblas_dgemm_strided_batched (N,N)", "rocblas_dgemm_strided_batched (T,N)", "rocblas_dsyrk_strided_batched",
         "rocblas_dsyrkx_strided_batched", "k_symm (variant by size)"]
print("executed flop: 2 K p^3 for the GEMMs, K p^2 (p + 1) for one triangle; useful flop of a symmetric product: K p^3")
for (K, p) in ((32, 500), (32, 512), (128, 500), (128, 512), (32, 1000), (32, 1024), (20, 200), (4, 500), (16, 500)):
    for mode in range(5):
        ms = np.zeros(1)
        rc = lib.ggl_dev_vendor_bench(K, p, mode, 20, ptr(ms))
        if rc != 0:
            print(f"K={K:3d} p={p:4d} {names[mode]:40s} failed: {lib.ggl_last_error().decode()}", flush=True)
            continue
        ex = (2.0 if mode < 2 else 1.0 + 1.0 / p) * K * p ** 3
        print(f"K={K:3d} p={p:4d} {names[mode]:40s} {ms[0] * 1e3:9.1f} us   executed {ex / (ms[0] * 1e-3) / 1e12:6.2f} TF/s   "
              f"as a symmetric product (K p^3 / t) {K * p ** 3 / (ms[0] * 1e-3) / 1e12:6.2f} TF/s", flush=True)
