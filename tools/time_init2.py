#!/usr/bin/env python3
"""ggl_ctx_create after a destroyed ctx of the same shape (arena pool hit), as the batch drivers produce it (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, batch, solver
p, n = 64, 100
S, _ = synth.make_problem("GGL", 1, p, N=2 * p, seed=1235)
lam = np.logspace(0, -2, n)
for rep in range(4):
    t0 = time.perf_counter()
    res = batch.ADMM_SGL_batch(S[0], lam, Omega_0=np.eye(p), X_0=np.eye(p), tol=1e-7, rtol=1e-7)
    print(rep, round((time.perf_counter() - t0) * 1e3, 2), "ms")
