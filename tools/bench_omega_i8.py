#!/usr/bin/env python3
"""VERDICT r3 item 3 -- the Omega-step on the int8 matrix cores (csrc/gemm_i8.hip: i8_omega_plan / i8_omega_run), stand-alone:
accuracy against numpy.linalg.eigh's phiplus and time per step against the FP64-MFMA chain's products at the same size.
The W stack is an ADMM iterate of the headline problem (iteration `it` of a CPU solve at small K, tiled to K instances).

    python tools/bench_omega_i8.py [p] [K ...]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib, synth
from gglasso_amd._lib import ptr
from oracle import ggl_oracle as orc


def iterate_W(p, Kgen, iters, seed=1239):
    """W = Theta - X - S / rho after `iters` ADMM iterations of the GGL problem (oracle), rho = 1."""
    S, _ = synth.make_problem("GGL", Kgen, p, N=2 * p, seed=seed)
    Om = np.repeat(np.eye(p)[None], Kgen, axis=0)
    Th, X = Om.copy(), np.zeros_like(Om)
    for _ in range(iters):
        W = Th - X - S
        Om, _ = orc.phiplus_stack(W, 1.0)
        Th = orc.prox_p(Om + X, 0.05, 0.01, "GGL")
        X = X + Om - Th
    return Th - X - S


def main():
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    Ks = [int(v) for v in sys.argv[2:]] or [4, 16, 32]
    lib = _lib.load_dev()
    Kgen = 4
    Wg = iterate_W(p, Kgen, 8)
    ref, _ = orc.phiplus_stack(Wg, 1.0)
    lam = np.array([np.linalg.eigvalsh(Wg[k] @ Wg[k] + 4 * np.eye(p))[-1] for k in range(Kgen)])
    print(f"p = {p}; W = ADMM iterate 8 of the headline problem; lambda_max(A') = {lam.round(2)}; |Omega|_max = {np.abs(ref).max():.3f}")
    for K in Ks:
        W = np.ascontiguousarray(np.stack([Wg[k % Kgen] for k in range(K)]))
        beta = np.ones(K)
        cb = np.array([lam[k % Kgen] * 1.02 for k in range(K)])
        ms64 = np.zeros(1)
        lib.ggl_dev_symm_bench(K, p, -1, 20, ptr(ms64))
        print(f"K={K:3d}  one FP64 product launch {ms64[0] * 1e3:6.1f} us (x7 = {7 * ms64[0] * 1e3:6.1f} us in isolation; the solver's "
              f"two overlapped chains run the 7 products of (32,500) in ~670 us)")
        cfgs = ((7, 4, 3, 5, 4),) if os.environ.get("I8_ONE_CFG") else ((7, 4, 3, 5, 4), (7, 5, 3, 5, 4), (7, 4, 3, 6, 5), (6, 4, 3, 5, 4), (8, 5, 3, 6, 5))
        for cfg in cfgs:
            for tol in (2e-12,):
                Om = np.zeros_like(W)
                out = np.zeros(4)
                c5 = (ctypes.c_int * 5)(*cfg)
                try:
                    _lib.check(lib.ggl_dev_omega_i8(K, p, ptr(W), ptr(beta), ptr(cb), c5, tol, ptr(Om), 10, ptr(out)))
                except Exception as e:  # noqa: BLE001
                    print(f"   cfg {cfg}: {e}")
                    continue
                err = max(np.abs(Om[k] - ref[k % Kgen]).max() for k in range(K))
                sym = max(np.abs(Om[k] - Om[k].T).max() for k in range(K))
                print(f"   cfg {cfg} tol {tol:g}: {out[0] * 1e3:7.1f} us per Omega-step ({int(out[1])} launches standing for {int(out[3])} "
                      f"fp64 products: {int(out[3]) * K * p ** 3 / (out[0] * 1e-3) / 1e12:5.1f} TF/s-equivalent)  max|dOmega| {err:.2e}  "
                      f"asym {sym:.1e} overflow={int(out[2])}", flush=True)


if __name__ == "__main__":
    main()
