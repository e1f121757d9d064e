#!/usr/bin/env python3
"""(Note: this tool's wrappers keep engines alive past their close(), so from the second repetition on every ctx creates
fresh streams -- ~6 ms each, "__init__ 22 ms" -- which plain repeated calls do not: tools/time_init2.py.)
Where a batched grid's wall time goes (dev tool): the C loop (HipEngine.batch_run), compactions (subset), the final download
of the snapshot stacks, the rest.   tools/time_batch.py [--p 1000] [--points 20] [--compact 0|1]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, batch, solver

ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=1000)
ap.add_argument("--points", type=int, default=20)
ap.add_argument("--compact", type=int, default=1)
a = ap.parse_args()
S, _ = synth.make_problem("GGL", 1, a.p, N=2 * a.p, seed=1235)
S = S[0]
lam = np.logspace(0, -2, a.points)
eye = np.eye(a.p)
T = {}
def timed(name, fn):
    def w(*args, **kw):
        t0 = time.perf_counter()
        r = fn(*args, **kw)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        T[name + "_calls"] = T.get(name + "_calls", 0) + 1
        return r
    return w
E = solver.HipEngine
from gglasso_amd import _lib as _L
_lib = _L.load()
for nm in ("ggl_ctx_create", "ggl_set_S_ex", "ggl_set_state_ex", "ggl_ctx_set_option", "ggl_ctx_destroy"):
    setattr(_lib, nm, timed("C:" + nm, getattr(_lib, nm)))
for nm in ("batch_run", "subset", "snapshots", "snapshot_state_from", "close", "__init__", "finalize_L", "selection_stats"):
    setattr(E, nm, timed(nm, getattr(E, nm)))
batch.ADMM_SGL_batch(S, lam[:2], Omega_0=eye, X_0=eye, max_iter=3)
import contextlib, io
calls = []
_br = E.batch_run
def br(self, n_iters, *args, **kw):
    t0 = time.perf_counter()
    k = _br(self, n_iters, *args, **kw)
    calls.append((self.K, k, round((time.perf_counter() - t0) * 1e3, 2)))
    return k
E.batch_run = br
for rep in range(4):
    T.clear(); calls.clear()
    res = None
    verbose = rep == 3                     # (the drivers' Python loop, for comparison)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        res = batch.ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=1e-7, rtol=1e-7, compact=bool(a.compact), verbose=verbose)
    tot = time.perf_counter() - t0
    print("python loop" if verbose else "C loop", "calls (K, iterations, ms):", calls, "carried", sum(i['carried'] for _, i in res))
    print(f"total {tot * 1e3:7.2f} ms | " + " ".join(f"{k} {v * 1e3:.2f}" if not k.endswith("_calls") else f"(x{v})" for k, v in T.items()))
