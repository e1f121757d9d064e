#!/usr/bin/env python3
"""The lambda1 x lambda2 model-selection grid of a multiple-graph problem solved as ONE batch on the GPU
(gglasso_amd.model_selection.grid_search -> batch.ADMM_MGL_batch: G grid points x K instances in one ctx), beside the
same grid walked point by point on the GPU with the reference's warm start (what a plain solver swap in the reference's
grid_search gives).  Default: the 8 x 1 grid at K = 4, p = 500 of VERDICT r1 item 3 -- 32 matrices per Omega-step, the
headline's batch, instead of eight under-filled solves.  Prints one JSON line.
    tools/bench_mgl_grid.py [--K 4] [--p 500] [--l1 8] [--l2 1] [--reg GGL]"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, model_selection as ms, batch, solver  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=4)
    ap.add_argument("--p", type=int, default=500)
    ap.add_argument("--l1", type=int, default=8)
    ap.add_argument("--l2", type=int, default=1)
    ap.add_argument("--reg", default="GGL")
    ap.add_argument("--tol", type=float, default=1e-7)
    a = ap.parse_args()
    K, p, N = a.K, a.p, 2 * a.p
    S, _ = synth.make_problem(a.reg, K, p, N=N, seed=1239)
    l1 = np.logspace(-0.7, -1.7, a.l1)
    l2 = np.logspace(-1.5, -2.5, a.l2)
    Nk = np.full(K, N)
    L1, L2 = ms.lambda_grid(l1, l2)
    lam1, lam2 = L1.T.ravel(), L2.T.ravel()
    quiet(batch.ADMM_MGL_batch, S, lam1[:2], lam2[:2], a.reg, max_iter=3, selection_stats=True)   # warm-up (library load: HIP code objects, rocSOLVER behind the criteria)
    t0 = time.perf_counter()
    res = quiet(batch.ADMM_MGL_batch, S, lam1, lam2, a.reg, tol=a.tol, rtol=a.tol)
    t_batch = time.perf_counter() - t0
    # with / without compaction of finished points, best of 3 each, interleaved
    t_c, t_u = [t_batch], []
    for _ in range(3):
        t0 = time.perf_counter()
        ru = quiet(batch.ADMM_MGL_batch, S, lam1, lam2, a.reg, tol=a.tol, rtol=a.tol, compact=False)
        t_u.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        quiet(batch.ADMM_MGL_batch, S, lam1, lam2, a.reg, tol=a.tol, rtol=a.tol)
        t_c.append(time.perf_counter() - t0)
    t_batch = min(t_c)
    its = [info['iterations'] for _, info in res]
    t0 = time.perf_counter()
    stats, ix, best = quiet(ms.grid_search, solver.ADMM_MGL, S, Nk, p, a.reg, l1, l2=l2, tol=a.tol, rtol=a.tol)
    t_total = time.perf_counter() - t0
    t0 = time.perf_counter()
    stats_s, ix_s, _ = quiet(ms.grid_search, solver.ADMM_MGL, S, Nk, p, a.reg, l1, l2=l2, tol=a.tol, rtol=a.tol, batched=False)
    t_seq = time.perf_counter() - t0
    out = {"workload": f"{a.reg} K={K} p={p}, {a.l1} x {a.l2} (lambda1, lambda2) grid, N={N}, tol=rtol={a.tol}",
           "batched_solve_s": t_batch, "batched_iterations_max": int(max(its)), "grid_point_iterations": int(sum(its)),
           "grid_point_iterations_per_s": sum(its) / t_batch, "batch_iterations_per_s": max(its) / t_batch,
           "executed_grid_point_iterations": int(sum(i['carried'] for _, i in res)),
           "uncompacted": {"batched_solve_s": min(t_u), "executed_grid_point_iterations": int(sum(i['carried'] for _, i in ru))},
           "grid_search_total_s": t_total, "criteria_and_download_s": t_total - t_batch,
           "sequential_warm_start_grid_search_s": t_seq, "speedup_vs_sequential": t_seq / t_total,
           "same_selection": [int(v) for v in ix] == [int(v) for v in ix_s],
           "best": {k: float(v) for k, v in stats['BEST'].items()}, "statuses": sorted({i['status'] for _, i in res})}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
