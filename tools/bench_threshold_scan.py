"""Round 3: tune_threshold (helper/model_selection.py:707-737) for every point of a lambda1 grid -- the 20 candidate
thresholds scored on the GPU by ``ggl_threshold_scan`` (distinct thresholded matrices only, K per eigenvalue launch)
against the host loop (eigvalsh + slogdet per threshold).  Also the rank statistic of a latent grid.
  python tools/bench_threshold_scan.py [p] [n_lambda]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, model_selection as ms
from gglasso_amd.batch import ADMM_SGL_batch

p = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = 2 * p
S, _ = synth.make_problem("GGL", 1, p, seed=11)
S = S[0]
lam = np.logspace(-0.5, -1.5, nl)
eye = np.eye(p)
taus = ms.default_tau_range()
ADMM_SGL_batch(S, lam[:2], Omega_0=eye, X_0=eye, tol=1e-7, rtol=1e-7, max_iter=3)       # warm the library up
t0 = time.perf_counter()
res0 = ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=1e-7, rtol=1e-7, selection_stats=True)
t_plain = time.perf_counter() - t0
t0 = time.perf_counter()
res = ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=1e-7, rtol=1e-7, selection_stats=True, tau_range=taus)
t_scan = time.perf_counter() - t0
n_eig = res[0][1]['selection']['threshold_eig_problems']
t0 = time.perf_counter()
same = True
for j in range(nl):
    Th = res[j][0]['Theta']
    _, tau, _ = ms.tune_threshold(Th, S, N, method='eBIC', gamma=0.3)
    jt = ms._pick_threshold(res[j][1]['selection']['threshold'], N, p, 'eBIC', 0.3)
    same &= bool(tau == taus[jt])
t_host = time.perf_counter() - t0
print(json.dumps({"workload": f"SGL p={p}, {nl}-point lambda1 grid, 20 thresholds per point (tune_threshold)",
                  "batched_solve_with_statistics_s": t_plain, "same_with_threshold_scan_s": t_scan,
                  "threshold_scan_s": t_scan - t_plain, "eigenvalue_problems": n_eig, "of_candidates": nl * len(taus),
                  "host_tune_threshold_s": t_host, "same_thresholds_chosen": same}))
