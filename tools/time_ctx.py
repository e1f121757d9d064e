#!/usr/bin/env python3
"""What one ADMM_MGL CALL costs around its iterations: ctx creation (allocations, streams, events), upload, exit checks,
download, ctx destruction -- the overheads a point-by-point grid walk (the reference's grid_search with gglasso_amd.ADMM_MGL as
its solver, seam 1) pays per grid point.

    python tools/time_ctx.py
"""
import contextlib
import ctypes
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib, solver, synth, ADMM_MGL

lib = _lib.load()
import sys as _s
CASES = [tuple(int(v) for v in a.split(':')) for a in _s.argv[1:]] or [(4, 500), (32, 500), (20, 200), (64, 100), (3, 50)]
for K, p in CASES:
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=1)
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    Z = np.zeros_like(S)
    eng = solver.HipEngine(S, eye, eye, Z)       # (first use of this size: code objects, handles)
    eng.step(1.0, 0.05, 0.01, "GGL", False, None, np.ones(K))
    eng.close()
    tc, tu, td = [], [], []
    for _ in range(5):
        t0 = time.perf_counter()
        h = _lib._vp()
        _lib.check(lib.ggl_ctx_create(0, K, p, 0, None, h))
        t1 = time.perf_counter()
        _lib.check(lib.ggl_set_S(h, _lib.ptr(S)))
        _lib.check(lib.ggl_set_state(h, _lib.ptr(eye), _lib.ptr(eye), None, _lib.ptr(Z)))
        t2 = time.perf_counter()
        lib.ggl_ctx_destroy(h)
        t3 = time.perf_counter()
        tc.append(t1 - t0); tu.append(t2 - t1); td.append(t3 - t2)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ADMM_MGL(S, 0.05, 0.01, "GGL", eye, tol=1e-7, rtol=1e-7)
        t0 = time.perf_counter()
        sol, info = ADMM_MGL(S, 0.05, 0.01, "GGL", eye, tol=1e-7, rtol=1e-7)
        t_call = time.perf_counter() - t0
        t0 = time.perf_counter()
        sol, info = ADMM_MGL(S, 0.05, 0.01, "GGL", eye, tol=1e-7, rtol=1e-7, measure=True)
        t_meas = time.perf_counter() - t0
    iters = len(info['residual'])
    print(f"K={K:3d} p={p:4d}: ctx create {np.median(tc) * 1e3:7.2f} ms | upload S + state {np.median(tu) * 1e3:7.2f} ms | destroy "
          f"{np.median(td) * 1e3:6.2f} ms || whole ADMM_MGL call {t_call * 1e3:7.1f} ms for {iters} iterations "
          f"({info['runtime'].sum() * 1e3:6.1f} ms inside the loop); with measure=True (objective every iteration) {t_meas * 1e3:7.1f} ms",
          flush=True)
