#!/usr/bin/env python3
"""Randomised pin of the oracle (runs ONLY in the build container, where /root/reference exists): the case generators of
tests/fuzz_checks.py -- the ones the GPU suite runs the product against the oracle with -- here run the REAL reference
(fabian-sp/GGLasso v0.2.1, imported from /root/reference/src as tests/golden/make_golden.py does) against the oracle: ADMM_MGL /
ADMM_SGL (solver/admm_solver.py:13-313, solver/single_admm_solver.py:15-275), block_SGL (:326-475), ext_ADMM_MGL
(solver/ext_admm_solver.py:18-323) and the operators (solver/ggl_helper.py) at drawn shapes and parameters.  Nothing of the
reference is written into the repo; the output of a run is kept in profiles/ as a record.

    python tests/golden/fuzz_oracle_vs_reference.py [cases per family] [seed]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
os.environ.setdefault("GGL_FUZZ_TOL", "1e-10")

import make_golden  # noqa: E402
admm, sadmm, gh, fh, dg, utils = make_golden._import_reference()
from gglasso.solver import ext_admm_solver  # noqa: E402

import fuzz_checks  # noqa: E402
import gglasso_amd.solver as prod_solver  # noqa: E402
import gglasso_amd.ext_solver as prod_ext  # noqa: E402
from oracle import ggl_oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0

# the generators call gglasso_amd.solver.* / gglasso_amd.ext_solver.* as "the thing under test": here that is the reference
prod_solver.ADMM_MGL = admm.ADMM_MGL
prod_solver.ADMM_SGL = sadmm.ADMM_SGL
prod_solver.block_SGL = sadmm.block_SGL
prod_ext.ext_ADMM_MGL = ext_admm_solver.ext_ADMM_MGL
fuzz_checks.P = [q for q in fuzz_checks.P if q <= 66]            # (the reference's Python loops: sizes it finishes in seconds)
fuzz_checks.PBIG = fuzz_checks.P
fuzz_checks.N_SAMPLES_INT = True

t0 = time.time()
total_bad = 0
for kind in ("solver", "block", "ext"):
    lines = []
    bad, notes, mx = fuzz_checks.run_cases(cases, seed, out=lines.append, kind=kind, big=False)
    total_bad += bad
    for ln in lines:
        print(ln[:400])
    print(f"{kind}: {cases} cases (seed {seed}), reference vs oracle: {bad} off by more than {fuzz_checks.TOL:g}; {notes} last-bit notes; "
          f"largest deviation of the rest {mx:.2e}; {time.time() - t0:.0f} s", flush=True)

# the operators: phiplus / prox_rank_norm from an eigendecomposition, prox_p, on the engineered spectra of fuzz_checks._spectrum
rng = np.random.default_rng(seed + 1000)
worst = {"phiplus": 0.0, "prox_rank_norm": 0.0, "prox_p": 0.0}
for i in range(cases):
    p = int(rng.choice([q for q in fuzz_checks.P if q >= 2]))
    Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
    W = (Q * fuzz_checks._spectrum(rng, p)) @ Q.T
    W = 0.5 * (W + W.T)
    beta = float(10.0 ** rng.uniform(-3, 1.5))
    D, V = np.linalg.eigh(W)
    ref = gh.phiplus(beta, D, V)
    mine, _ = orc.phiplus_stack(W[None], np.array([beta]))
    worst["phiplus"] = max(worst["phiplus"], float(np.abs(mine[0] - ref).max()) / max(1.0, float(np.abs(ref).max())))
    ref = gh.prox_rank_norm(W, beta, D, V)
    mine = orc.rank_stack(W[None], np.array([beta]))
    worst["prox_rank_norm"] = max(worst["prox_rank_norm"], float(np.abs(mine[0] - ref).max()) / max(1.0, float(np.abs(ref).max())))
    K = int(rng.integers(2, 9))
    X = rng.standard_normal((K, p, p))
    if rng.random() < 0.5:
        X = np.round(X * 2) / 2                               # ties across K and exact zeros
    X = 0.5 * (X + X.transpose(0, 2, 1))
    l1, l2 = float(10.0 ** rng.uniform(-3, 0.5)), float(10.0 ** rng.uniform(-3, 0.5))
    for reg in ("GGL", "FGL"):
        ref = gh.prox_p(X.copy(), l1, l2, reg)
        mine = orc.prox_p(X.copy(), l1, l2, reg)
        worst["prox_p"] = max(worst["prox_p"], float(np.abs(mine - ref).max()))
print(f"operators: {cases} cases: largest deviation oracle vs reference {worst}; {time.time() - t0:.0f} s")
total_bad += sum(v > 1e-10 for v in worst.values())
print("ok" if total_bad == 0 else f"{total_bad} findings")
