#!/usr/bin/env python3
"""One-off robustness sweep (dev tool): random full solves through the HIP path vs the CPU oracle.
Sizes above the Jacobi limit so that the Newton-Schulz Omega-/L-steps, speculation and the poll-wait are all in play."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, solver
from oracle import ggl_oracle as orc

rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
worst = 0.0
n = int(os.environ.get("N", "16"))
for t in range(n):
    reg = ("GGL", "FGL")[t % 2]
    latent = bool((t // 2) % 2)
    K = int(rng.integers(2, 7))
    p = int(rng.integers(130, 240))
    l1 = float(10 ** rng.uniform(-2, -0.7))
    l2 = float(10 ** rng.uniform(-2.5, -1))
    rho = float(10 ** rng.uniform(-1, 1))
    S, _ = synth.make_problem(reg, K, p, N=2 * p, seed=int(rng.integers(1 << 30)))
    Om0 = np.stack([np.eye(p)] * K)
    kw = dict(tol=1e-8, rtol=1e-8, rho=rho, latent=latent, mu1=0.2 if latent else None, max_iter=400)
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.time(); ref, ri = orc.ADMM_MGL(S, l1, l2, reg, Om0, **kw); tc = time.time() - t0
        t0 = time.time(); sol, si = solver.ADMM_MGL(S, l1, l2, reg, Om0, **kw); tg = time.time() - t0
    err = np.linalg.norm(sol['Theta'] - ref['Theta'])
    worst = max(worst, err)
    ok = (si['status'] == ri['status']) and err <= 1e-8
    print(f"{t:2d} {reg} K={K} p={p} latent={latent} l1={l1:.3g} l2={l2:.3g} rho0={rho:.3g}: status {si['status']!r} "
          f"|dTheta|_F {err:.2e}  cpu {tc:.1f}s gpu {tg:.2f}s  {'ok' if ok else 'MISMATCH'}", flush=True)
print("worst", worst)
