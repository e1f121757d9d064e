#!/usr/bin/env python3
"""Where the FGL Theta-step's time goes (GGL_DEV build; GGL_FGL_ABL = 1 no Condat scan, 2 neither scan nor soft-threshold,
3 scan only, 4 the nested-loop scan that ran until round 3): Theta phase of the non-latent FGL iteration, HIP events.   python tools/bench_fgl_theta.py [K p]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
_lib.LIB_PATH = _lib.DEV_LIB_PATH          # the dev build carries the ablation instances
from gglasso_amd import synth, solver

K, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 500)
S, _ = synth.make_problem("FGL", K, p, N=2 * p, seed=1237)
Om0 = np.repeat(np.eye(p)[None], K, axis=0)
eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S))
nk = np.ones(K)
for _ in range(12):
    eng.step(1.0, 0.05, 0.01, "FGL", False, None, nk)
eng.save_state()
for abl in ("0", "4", "1", "2", "3"):
    if abl == "0":
        os.environ.pop("GGL_FGL_ABL", None)
    else:
        os.environ["GGL_FGL_ABL"] = abl
    eng.restore_state()
    eng.profile(1)
    eng.profile_read(reset=True)
    for _ in range(6):
        eng.step(1.0, 0.05, 0.01, "FGL", False, None, nk)
    ph = eng.profile_read(reset=True)
    ms, n = ph["theta"]
    B = 8.0 * K * p * p
    print(f"K={K} p={p} ablation {abl}: theta {ms / n * 1e3:8.1f} us per launch   ({5 * B / (ms / n * 1e-3) / 1e12:.2f} TB/s on 5B)", flush=True)
eng.close()
