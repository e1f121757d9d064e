import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gglasso_amd import synth, solver
for (K, p) in ((32, 500), (20, 200), (4, 500)):
    S, _ = synth.make_problem("GGL", K, p, seed=1239)
    Om0 = np.stack([np.eye(p)] * K)
    eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S))
    nk = np.ones(K)
    for _ in range(5): eng.step(1.0, 0.05, 0.01, "GGL", False, None, nk)
    t0 = time.perf_counter()
    for _ in range(50): eng.step(1.0, 0.05, 0.01, "GGL", False, None, nk)
    t1 = time.perf_counter()
    eng.profile(True); eng.profile_read()
    for _ in range(50): eng.step(1.0, 0.05, 0.01, "GGL", False, None, nk)
    pr = eng.profile_read()
    gpu = sum(ms for ms, cnt in pr.values()) / 50
    print(f"K={K} p={p}: step-only loop {1e3*(t1-t0)/50:.4f} ms/iter; sum of GPU phases {gpu:.4f} ms/iter", flush=True)
    eng.close()
