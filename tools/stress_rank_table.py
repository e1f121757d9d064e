#!/usr/bin/env python3
"""Repeats the latent grid of tests/test_gpu_selection.py::test_rank_of_the_solvers_latent_component (p = 500) and reports every
run whose RANK table differs from the eigendecomposition route's -- the test failed once in a full-suite run of round 5 and
never alone.  Between the repetitions ctxs of other shapes are created, stepped and destroyed so that pooled arenas and streams
change hands (the same arena sizes as the grid's ctx and as its compacted subsets among them).
    python tools/stress_rank_table.py [repetitions] [p] [poison byte: 0 zeros, 1 = 0xFF, 127 = 0x7F, 71 = 0x47]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import solver, synth, batch, _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
poison = int(sys.argv[3]) if len(sys.argv) > 3 else int(os.environ.get("GGL_DEBUG_POISON", "0") or 0)
S, _ = synth.make_problem("GGL", 1, p, seed=3)
lam, mu = np.array([0.1, 0.2]), np.array([0.5, 1.0, 2.0])
lam6, mu6 = np.repeat(lam, 3), np.tile(mu, 2)
eye = np.eye(p)


def grid(**kw):
    """The batch single_grid_search runs, with everything a failure would want to know."""
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        res = batch.ADMM_SGL_batch(S[0], lam6, Omega_0=eye, X_0=eye, tol=1e-8, rtol=1e-8, latent=True, mu1=mu6,
                                   selection_stats=True, **kw)
    ranks = np.array([info['selection']['rank'] for _, info in res]).reshape(2, 3)
    L = np.stack([sol['L'] for sol, _ in res])
    meta = dict(status=[info['status'] for _, info in res], iters=[info['iterations'] for _, info in res],
                carried=[info['carried'] for _, info in res], errors=[info.get('error') for _, info in res],
                warnings=[str(w.message) for w in wl], nan=[int(np.isnan(A).sum()) for A in L],
                normL=[float(np.linalg.norm(A)) for A in L])
    return ranks, L, meta


def churn(r):
    """ctxs of other shapes in between: same arena size as the grid's ctx (K = 6) with GGL data, the sizes of its compacted
    subsets (K = 4, 3, 2), a batch of small problems, and now and then a big slab (its arena is not pooled, its streams are)."""
    kinds = [("GGL", 6, p), ("GGL", 4, p), ("FGL", 3, p), ("GGL", 2, p), ("GGL", 40, 48), ("GGL", 8, 1000)]
    reg, K, q = kinds[r % len(kinds)]
    Sb, _ = synth.make_problem(reg, K, q, seed=5 + r)
    I = np.stack([np.eye(q)] * K)
    eng = solver.HipEngine(Sb, I, I, np.zeros_like(Sb))
    lat = (r % 4 == 2)
    for _ in range(3):
        eng.step(1.0, 0.05, 0.01, reg, lat, np.full(K, 0.3) if lat else None, np.ones(K))
    eng.close()


if poison:
    _lib.load().ggl_debug_poison(poison)
saved = dict(solver.ENGINE_OPTIONS)
solver.ENGINE_OPTIONS["rank_eig"] = 1.0
want, low_e, meta_e = grid()
solver.ENGINE_OPTIONS.clear()
solver.ENGINE_OPTIONS.update(saved)
want_np = np.array([np.linalg.matrix_rank(A, hermitian=True) for A in low_e]).reshape(2, 3)
print(f"poison byte {poison:#x}; eigendecomposition route: RANK {want.tolist()} (numpy on its L: {want_np.tolist()}) "
      f"iterations {meta_e['iters']} carried {meta_e['carried']} status {sorted(set(meta_e['status']))}", flush=True)
bad = 0
for r in range(reps):
    churn(r)
    t0 = time.perf_counter()
    ranks, L, meta = grid()
    dt = time.perf_counter() - t0
    ok = np.array_equal(ranks, want) and np.abs(L - low_e).max() <= 1e-7 and set(meta['status']) == {'optimal'} \
        and not meta['warnings']
    if not ok:
        bad += 1
        print(f"run {r}: RANK {ranks.tolist()}  max|L_ns - L_eig| {[float(np.abs(a - b).max()) for a, b in zip(L, low_e)]}\n"
              f"   {meta}  {dt:.2f} s", flush=True)
    elif r == 0:
        print(f"run 0 ok: iterations {meta['iters']} carried {meta['carried']}  {dt:.2f} s", flush=True)
print(f"{bad} of {reps} runs differ (poison byte {poison:#x})")
sys.exit(1 if bad else 0)
