#!/usr/bin/env python3
"""The reference's own published benchmark (BASELINE.md / SURVEY.md section 6: ADMM_SGL, tol = rtol = 1e-8, lambda1 = 0.05, N = 1.1 p,
AMD Opteron 6378; data/synthetic/bm2000.csv, bm5000.csv) on this library: whole gglasso_amd.ADMM_SGL calls (upload, iterations, exit
checks, download) on the same kind of problem (sparse power-law-free synthetic precision, gglasso_amd.synth), iterations and
iterations per second of the CALL.  Not the BASELINE metric -- context for a reader who knows the reference's numbers.

    python tools/bench_sgl_published.py [p ...]
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, ADMM_SGL

PUBLISHED = {100: (0.0868, 53), 500: (1.096, 33), 1000: (4.251, 27), 2000: (30.58, 25), 4000: (190.2, 18), 5000: (308.6, 16)}
for p in [int(v) for v in sys.argv[1:]] or [100, 500, 1000, 2000, 4000]:
    S, _ = synth.make_problem("GGL", 1, p, N=int(1.1 * p), seed=1250)
    S = S[0]
    eye = np.eye(p)
    with contextlib.redirect_stdout(io.StringIO()):
        ADMM_SGL(S, 0.05, eye, tol=1e-8, rtol=1e-8, max_iter=3)               # (code objects, handles)
        t0 = time.perf_counter()
        sol, info = ADMM_SGL(S, 0.05, eye, tol=1e-8, rtol=1e-8, measure=True)
        dt = time.perf_counter() - t0
    it = len(info['residual'])
    pub = PUBLISHED.get(p)
    print(f"p={p:5d}: {info['status']}, {it} iterations, whole call {dt:8.3f} s = {it / dt:8.1f} it/s (in the loop {info['runtime'].sum():7.3f} s)"
          + (f"   | reference, published (Opteron 6378): {pub[0]} s / {pub[1]} iterations = {pub[1] / pub[0]:.2f} it/s" if pub else ""),
          flush=True)
