#!/usr/bin/env python3
"""Wall-clock timing of ADMM iterations through the C ABI for a few (reg,K,p) shapes (dev tool)."""
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, solver, _lib  # noqa: E402


def run(reg, K, p, latent=False, eig=_lib.EIG_AUTO, iters=10, warm=2):
    S, _ = synth.make_problem(reg if reg != 'SGL' else 'GGL', K, p, seed=1)
    Om0 = np.stack([np.eye(p)] * K)
    eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S), eig=eig)
    mu1 = 0.5 * np.ones(K)
    nk = np.ones(K)
    rho = 1.0
    ts = []
    for it in range(warm + iters):
        t0 = time.perf_counter()
        sq = eng.step(rho, 0.05, 0.01, reg, latent, mu1 if latent else None, nk)
        ts.append(time.perf_counter() - t0)
    eng.close()
    t = np.array(ts[warm:])
    print(f"{reg} K={K} p={p} latent={latent} eig={eig}: median {np.median(t)*1e3:.3f} ms/iter  "
          f"min {t.min()*1e3:.3f}  ({1/np.median(t):.1f} it/s)  norms={sq[:2]}", flush=True)


if __name__ == "__main__":
    cfgs = [("GGL", 3, 50, False, 0), ("GGL", 20, 100, False, 1), ("GGL", 20, 100, False, 2),
            ("GGL", 20, 200, False, 0), ("GGL", 32, 500, False, 0), ("FGL", 50, 500, True, 0),
            ("SGL", 1, 1000, False, 0), ("GGL", 32, 1000, False, 0)]
    for c in cfgs:
        try:
            run(*c)
        except Exception as e:  # noqa: BLE001
            print("FAILED", c, e, flush=True)
