#!/usr/bin/env python3
"""C2 of SURVEY.md section 8: the lambda1 grid of a Single Graphical Lasso problem (p = 1000, 20 points,
logspace(0,-2,20)) solved as ONE batch on the GPU (gglasso_amd.model_selection.single_grid_search), beside the
same grid walked point by point on the GPU with the reference's warm start (what a plain solver swap in
single_grid_search gives).  Prints one JSON line.   tools/bench_grid.py [--p 1000] [--points 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, model_selection as ms, batch, solver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=int, default=1000)
    ap.add_argument("--points", type=int, default=20)
    ap.add_argument("--tol", type=float, default=1e-7)
    ap.add_argument("--no-sequential", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="ctx option for every engine")
    a = ap.parse_args()
    solver.ENGINE_OPTIONS.update({kv.split("=")[0]: float(kv.split("=")[1]) for kv in a.opt})
    p, N = a.p, 2 * a.p
    S, _ = synth.make_problem("GGL", 1, p, N=N, seed=1235)
    S = S[0]
    lam = np.logspace(0, -2, a.points)
    eye = np.eye(p)
    batch.ADMM_SGL_batch(S, lam[:2], Omega_0=eye, X_0=eye, max_iter=3, selection_stats=True)   # warm-up (library load: HIP code objects, rocSOLVER behind the criteria)
    t0 = time.perf_counter()
    res = batch.ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=a.tol, rtol=a.tol)
    t_batch = time.perf_counter() - t0
    its = [info['iterations'] for _, info in res]
    carried = [info['carried'] for _, info in res]
    # the same batch without compaction of finished points: best of 3 each, interleaved
    t_c, t_u = [t_batch], []
    for _ in range(3):
        t0 = time.perf_counter()
        ru = batch.ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=a.tol, rtol=a.tol, compact=False)
        t_u.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        batch.ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=a.tol, rtol=a.tol)
        t_c.append(time.perf_counter() - t0)
    t_batch = min(t_c)
    t0 = time.perf_counter()
    best, est, _, st = ms.single_grid_search(S, lam, N, tol=a.tol, rtol=a.tol)
    t_total = time.perf_counter() - t0
    out = {"workload": f"SGL p={p}, {a.points}-point lambda1 grid logspace(0,-2), N={N}, tol=rtol={a.tol}",
           "batched_solve_s": t_batch, "batched_iterations_max": int(max(its)), "grid_point_iterations": int(sum(its)),
           "grid_point_iterations_per_s": sum(its) / t_batch, "batch_iterations_per_s": max(its) / t_batch,
           "executed_grid_point_iterations": int(sum(carried)),
           "uncompacted": {"batched_solve_s": min(t_u), "executed_grid_point_iterations": int(sum(i['carried'] for _, i in ru))},
           "single_grid_search_total_s": t_total, "criteria_and_download_s": t_total - t_batch,
           "best_lambda1": float(st['BEST']['lambda1']), "statuses": sorted({i['status'] for _, i in res})}
    if not a.no_sequential:
        t0 = time.perf_counter()
        Om0, seq_its = eye, 0
        for l1 in lam:
            sol, info = solver.ADMM_SGL(S, l1, Om0, X_0=eye, tol=a.tol, rtol=a.tol, verbose=False, measure=True)
            Om0 = sol['Omega']
            seq_its += len(info['residual'])
        out["sequential_warm_start_s"] = time.perf_counter() - t0
        out["sequential_iterations"] = int(seq_its)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
