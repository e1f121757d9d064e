#!/usr/bin/env python3
"""VERDICT r3 item 3b -- the symmetric product on the INT8 matrix cores (csrc/gemm_i8.hip) on the device: bit-for-bit against
the NumPy emulation of tools/proto_ozaki.py, and its time per launch against the FP64-MFMA product kernel at the same size.

    python tools/bench_i8.py [p] [K ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
import proto_ozaki as oz


def main():
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    Ks = [int(v) for v in sys.argv[2:]] or [4, 16, 32]
    lib = _lib.load_dev()
    rng = np.random.default_rng(5)
    print(f"p = {p}: product launch, microseconds (HIP events, mean of 20); 'eq TF/s' = K p^3 / time: what an FP64 product "
          f"kernel would have to run at to match (FP64 matrix peak 78.6)")
    for K in Ks:
        # commuting symmetric operands with spectrum in [-1, 1]: A random symmetric / |A|_2, B a polynomial in A
        A = np.empty((K, p, p))
        B = np.empty((K, p, p))
        for k in range(K):
            G = rng.standard_normal((p, p))
            G = 0.5 * (G + G.T)
            G /= np.linalg.norm(G, 2) * 1.0001
            A[k] = G
            B[k] = 0.5 * np.eye(p) + 0.3 * G - 0.2 * G @ G
            B[k] = 0.5 * (B[k] + B[k].T)
        ms = np.zeros(1)
        lib.ggl_dev_symm_bench(K, p, -1, 20, ptr(ms))
        t64 = ms[0] * 1e3
        row = [f"K={K:3d}  fp64 kernel {t64:7.1f} us ({K * p ** 3 / t64 / 1e6:5.1f} TF/s)"]
        for S, dmax in ((7, 6), (6, 5), (5, 4), (4, 3), (3, 2), (2, 1)):
            C = np.zeros((K, p, p))
            out = np.zeros(3)
            _lib.check(lib.ggl_dev_symm_i8(K, p, S, dmax, ptr(A), ptr(B), 1.0, 1.0, ptr(C), 20, ptr(out)))
            npairs = sum(1 for t in range(S) for u in range(S) if t + u <= dmax)
            # bit-for-bit against the emulation (instance 0), accuracy against the float64 product
            ref = oz.oz_mul(A[0], B[0], S, S, 1.0, 1.0, dmax)
            iu = np.triu_indices(p)
            exact = bool(np.array_equal(C[0][iu], ref[iu]))
            err = np.abs(C[0] - A[0] @ B[0])[iu].max()
            us = out[1] * 1e3
            row.append(f"  S={S} ({npairs:2d} pairs): {us:7.1f} us  {npairs * 2 * K * (64 * ((p + 63) // 64)) ** 3 / us / 1e6:7.0f} TOPS "
                       f"eq {K * p ** 3 / us / 1e6:6.1f} TF/s  slicing {out[0] * 1e3 / 2:5.1f} us/operand  "
                       f"bitwise={exact} err {err:.1e} overflow={int(out[2])}")
        print("\n".join(row), flush=True)


if __name__ == "__main__":
    main()
