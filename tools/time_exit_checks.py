#!/usr/bin/env python3
"""What the end of a solve costs beside its iterations: ggl_exit_checks (eigenvalues of Theta - L and of L: admm_solver.py:284-301),
ggl_finalize_L, and the down/upload -- against the time of the ADMM iterations themselves.

    python tools/time_exit_checks.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import solver, synth


def t(fn, n=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


for reg, K, p, latent in (("GGL", 32, 500, False), ("FGL", 50, 500, True), ("GGL", 64, 100, False), ("GGL", 20, 200, False),
                          ("GGL", 8, 1000, False), ("GGL", 3, 50, True)):
    S, _ = synth.make_problem(reg, K, p, N=2 * p, seed=1)
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    t0 = time.perf_counter()
    eng = solver.HipEngine(S, eye, eye, np.zeros_like(S))
    eng.sync()
    t_up = (time.perf_counter() - t0) * 1e3
    mu = 0.5 * np.ones(K) if latent else None
    for _ in range(5):
        eng.step(1.0, 0.05, 0.01, reg, latent, mu, np.ones(K))

    def it():
        eng.step(1.0, 0.05, 0.01, reg, latent, mu, np.ones(K))
    t_it = t(it, 10)
    t_ex = t(lambda: eng.exit_checks(latent))
    fast = getattr(eng, "exit_checks_fast", None)
    t_fast = t(lambda: fast(latent, 1e-5)) if fast else float("nan")
    t_st = t(lambda: eng.state())
    print(f"{reg} K={K} p={p} latent={latent}: ctx + upload {t_up:7.1f} ms | iteration {t_it:6.3f} ms | exit_checks {t_ex:7.2f} ms "
          f"| exit_checks_fast {t_fast:7.2f} ms | download {t_st:6.1f} ms", flush=True)
    eng.close()
