#!/usr/bin/env python3
"""Per-iteration profile of the headline solve from the identity start: wall time, rho, products of the Omega-step,
speculative / repeated / non-speculative, pre-launched chains dropped.   python tools/iter_profile.py [iters] [opt=val ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, solver

iters = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 60
opts = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
K, p = 32, 500
S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=1239)
Om0 = np.repeat(np.eye(p)[None], K, axis=0)
eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S), options=opts)
nk = np.ones(K)
dim = K * (p * p + p) / 2
eng.save_state()
for rep in range(3):
    if os.environ.get("HOST_RESET"):
        eng.set_state(Om0, Om0, np.zeros_like(S))
    else:
        eng.restore_state()
    rho = 1.0
    rows = []
    prev = eng.ns_stats()
    eng.profile(2)
    eng.profile_read(reset=True)
    eng.sync()
    for it in range(iters):
        t0 = time.perf_counter()
        sq = eng.step(rho, 0.05, 0.01, "GGL", False, None, nk)
        r, s, ep, ed = solver.residuals_from_norms(sq, rho, 1e-20, 1e-20, dim)
        rn = solver.next_rho(rho, r, s)
        if rn != rho:
            eng.scale_X(rho / rn)
        dt = time.perf_counter() - t0
        st = eng.ns_stats()
        ph = eng.profile_read(reset=True)
        eig = sum(ph[q][0] for q in ("eig_omega", "eig_omega2"))
        rows.append((it, dt * 1e3, eig, rho, st["units"] - prev["units"], st["calls"] - prev["calls"],
                     st["spec_calls"] - prev["spec_calls"], st["spec_misses"] - prev["spec_misses"],
                     st["pre_dropped"] - prev["pre_dropped"], r, s))
        prev = st
        rho = rn
print(" it     ms  eig_ms    rho  products calls spec miss dropped        r          s")
for row in rows:
    print("%3d %6.3f %7.3f %6.3f %8d %5d %4d %4d %7d %10.3e %10.3e" % row)
t = np.array([r[1] for r in rows])
print(f"iterations 5..24: {t[5:25].mean():.4f} ms/it   5..54: {t[5:55].mean():.4f}   25..54: {t[25:55].mean():.4f}")
eng.close()
