#!/usr/bin/env python3
"""The Omega-step of small matrices as ONE launch with the chain resident in LDS (csrc/omega_lds.hip), stand-alone:
accuracy against numpy.linalg.eigh's phiplus and time per launch, on ADMM iterates of a GGL problem of that size.

    python tools/bench_omega_lds.py [K:p ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib, synth
from gglasso_amd._lib import ptr
from oracle import ggl_oracle as orc


def iterate(p, K, iters, seed=1239, rho=1.0):
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=seed)
    Om = np.repeat(np.eye(p)[None], K, axis=0)
    Th, X = Om.copy(), np.zeros_like(Om)
    for _ in range(iters):
        Om, _ = orc.phiplus_stack(Th - X - S / rho, 1.0 / rho)
        Th = orc.prox_p(Om + X, 0.05 / rho, 0.01 / rho, "GGL")
        X = X + Om - Th
    return np.ascontiguousarray(Th), np.ascontiguousarray(X), np.ascontiguousarray(S)


def main():
    cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(256, 64), (256, 48), (256, 32), (256, 16), (64, 64), (1024, 64), (256, 50), (256, 9)]
    lib = _lib.load()
    for K, p in cases:
        Kg = min(K, 8)
        for it, rho in ((0, 1.0), (8, 1.0), (8, 4.0)):
            Th, X, S = iterate(p, Kg, it, rho=rho)
            rep = lambda a: np.ascontiguousarray(np.stack([a[k % Kg] for k in range(K)]))
            Th, X, S = rep(Th), rep(X), rep(S)
            beta = np.full(K, 1.0 / rho)
            W = Th - X - beta[:, None, None] * S
            ref, _ = orc.phiplus_stack(W, 1.0 / rho)
            for tol, waves in ((2e-12, 4), (2e-12, 8), (1e-10, 4), (1e-10, 8)):
                Om = np.zeros_like(W)
                cb = np.zeros(K)
                out = np.zeros(18)
                _lib.check(lib.ggl_dev_omega_lds(K, p, ptr(Th), None, ptr(X), ptr(S), ptr(beta), tol, 9 + 1000 * waves, ptr(Om), ptr(cb), 20, ptr(out)))
                lam = np.linalg.eigvalsh(W[0] @ W[0] + 4 * beta[0] * np.eye(p))[-1]
                err = np.abs(Om - ref).max() / np.abs(ref).max()
                sym = np.abs(Om - Om.transpose(0, 2, 1)).max()
                print(f"K={K:5d} p={p:3d} iterate {it} rho={rho}: tol {tol:7.0e} waves {waves}  {out[0] * 1e3:7.1f} us  flag {int(out[1])}  products/instance "
                      f"{out[2] / K:4.1f}  rel err {err:8.2e}  asym {sym:.1e}  bound/lambda_max {cb[0] / lam:6.3f}  kappa {cb[0] / (4 * beta[0]):6.1f}")
                if tol == 2e-12 and p > 32:
                    print("      instance 0 (us): form W %.2f | A' %.2f | B' %.2f | bound %.2f | first step %.2f | steps %.2f | (gap %.2f) W again %.2f | Omega %.2f | products %d"
                          % (out[5], out[6] - out[5], out[7] - out[6], out[8] - out[7], out[9] - out[8], out[10] - out[9], 0.0, out[11] - out[10], out[12] - out[11], out[13]))


if __name__ == "__main__":
    main()
