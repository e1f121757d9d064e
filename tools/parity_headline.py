#!/usr/bin/env python3
"""One-off end-to-end parity at the headline's regime (dev tool): GGL K=16/32, p=500 -- concurrent parts, direct-to-LDS
kernels, speculation -- converged solve through the HIP path vs the CPU oracle (minutes of CPU time).
    K=32 TOL=1e-10 python tools/parity_headline.py [name=value ...]      ctx options, e.g. ns_tol=1e-12; several values of
one option separated by commas are run in turn against the same oracle solve (ns_tol=4e-16,1e-13,1e-12)."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, solver
from oracle import ggl_oracle as orc
K, p = int(os.environ.get("K", "16")), int(os.environ.get("P", "500"))
REG = os.environ.get("REG", "GGL")               # REG=FGL LATENT=1 K=50: C4 (the two-tier L-step at full size)
LATENT = os.environ.get("LATENT", "0") == "1"
S, _ = synth.make_problem(REG, K, p, N=2 * p, seed=int(os.environ.get("SEED", "1239")))
Om0 = np.stack([np.eye(p)] * K)
kw = dict(tol=float(os.environ.get("TOL", "1e-7")), rtol=float(os.environ.get("TOL", "1e-7")), max_iter=300)
if LATENT:
    kw.update(latent=True, mu1=0.5 * np.ones(K))
sweeps = [{}]
for arg in sys.argv[1:]:
    name, values = arg.split("=")
    sweeps = [dict(s, **{name: float(v)}) for s in sweeps for v in values.split(",")]
with contextlib.redirect_stdout(io.StringIO()):
    t0 = time.time(); ref, ri = orc.ADMM_MGL(S, 0.05, 0.01, REG, Om0, **kw); tc = time.time() - t0
for opts in sweeps:
    solver.ENGINE_OPTIONS.clear()
    solver.ENGINE_OPTIONS.update(opts)
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.time(); sol, si = solver.ADMM_MGL(S, 0.05, 0.01, REG, Om0, measure=True, **kw); tg = time.time() - t0
    nT = np.linalg.norm(ref['Theta'])
    print(f"{REG}{' latent' if LATENT else ''} K={K} p={p} options {opts}: status {si['status']!r}/{ri['status']!r} iterations {len(si['residual'])}/{ri['iterations']} "
          f"|dTheta|_F {np.linalg.norm(sol['Theta'] - ref['Theta']):.3e} (|Theta|_F {nT:.3e}) max|dTheta| {np.abs(sol['Theta'] - ref['Theta']).max():.3e} "
          f"max|dOmega| {np.abs(sol['Omega'] - ref['Omega']).max():.3e}" + (f" max|dL| {np.abs(sol['L'] - ref['L']).max():.3e}" if LATENT else "") + f" gpu {tg:.2f}s cpu {tc:.1f}s", flush=True)
