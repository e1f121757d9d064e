#!/usr/bin/env python3
"""VERDICT r3 item 5 -- the one-workgroup-per-matrix LDS Jacobi eigensolver (csrc/eig_jacobi.hip) measured: per ADMM iteration
of a GGL solve at p <= 128, the Omega phase's time (HIP events; the kernel fuses eigh + phiplus + reconstruction), the sweeps
every instance took, and what that is against the kernel's own rooflines:
  flop   one sweep = p(p-1)/2 rotations x (3 dot products + 1 rotation of two p-vectors) = 6 p + 6 p flop -> 6 p^3 per sweep
  LDS    one sweep reads and writes both rows of every pair: p(p-1)/2 x 4 p x 8 B = 16 p^3 bytes per sweep and matrix
and the crossover against the Newton-Schulz route (GGL_EIG_NEWTON_SCHULZ forces the matrix-function Omega-step below 128).

    python tools/bench_jacobi.py [K:p ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib, solver, synth


def run(K, p, eig, iters=30, warm=None):
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=1241)
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    opts = {} if warm is None else {"jacobi_warm": warm}
    eng = solver.HipEngine(S, eye, eye, np.zeros_like(S), eig=eig, options=opts)
    try:
        rho, nk = 1.0, np.ones(K)
        sweeps, ms = [], []
        for it in range(iters):
            eng.profile(1)
            sq = eng.step(rho, 0.05, 0.01, "GGL", False, None, nk)
            ph = eng.profile_read()
            ms.append(sum(ph[k][0] for k in ("eig_omega", "recon_omega", "eig_omega2") if k in ph))
            if eig != _lib.EIG_NEWTON_SCHULZ:
                sweeps.append(eng.eig_info().copy())
            r_t, s_t, _, _ = solver.residuals_from_norms(sq, rho, 1e-20, 1e-20, K * (p * p + p) / 2)
            rn = solver.next_rho(rho, r_t, s_t)
            if rn != rho:
                eng.scale_X(rho / rn)
                rho = rn
        return np.array(ms), (np.array(sweeps) if sweeps else None)
    finally:
        eng.close()


def main():
    cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(64, 100), (256, 64), (32, 128), (512, 32), (20, 50)]
    for K, p in cases:
        ms, sw = run(K, p, _lib.EIG_JACOBI)
        late = slice(10, None)
        t = ms[late].mean() * 1e-3
        s_mean = sw[late].mean()
        flop = 6.0 * p ** 3 * s_mean * K
        lds = 16.0 * p ** 3 * s_mean * K
        print(f"K={K:4d} p={p:4d}  Jacobi: Omega phase {ms[late].mean() * 1e3:8.1f} us  sweeps/instance: first iteration {sw[0].mean():.1f}, "
              f"iterations 10+ mean {s_mean:.2f} (min {sw[late].min()}, max {sw[late].max()})  "
              f"{flop / t / 1e12:6.2f} TF/s of rotations, {lds / t / 1e12:6.2f} TB/s of LDS traffic "
              f"({min(K, 256)} CUs busy: {lds / t / 1e9 / min(K, 256):6.1f} GB/s per CU)", flush=True)
        try:
            ms2, _ = run(K, p, _lib.EIG_NEWTON_SCHULZ)
            print(f"                 Newton-Schulz route (FP64 MFMA products): Omega phase {ms2[late].mean() * 1e3:8.1f} us", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"                 Newton-Schulz route: {e}")


if __name__ == "__main__":
    main()
