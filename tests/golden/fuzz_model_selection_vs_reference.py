#!/usr/bin/env python3
"""Randomised pin of the model-selection HOST logic (runs ONLY in the build container, where /root/reference exists): the
reference's grid_search and single_grid_search (helper/model_selection.py:55-298, 505-692) against gglasso_amd.model_selection's
sequential walk -- the branch that takes any solver callable (``batched=False``) resp. the point-by-point branch of
single_grid_search -- with the REFERENCE's own solver on both sides, so that what is compared is the walk, the criteria, the
thresholding and the selection, on random grids the fixtures G12 / G16 / G17 do not hold.  No GPU involved.

    python tests/golden/fuzz_model_selection_vs_reference.py [cases] [seed]"""
import contextlib
import io
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden  # noqa: E402
admm, sadmm, gh, fh, dg, utils = make_golden._import_reference()
from gglasso.helper import model_selection as rms  # noqa: E402

from gglasso_amd import model_selection as ms, synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ms.ADMM_SGL = sadmm.ADMM_SGL                       # the point-by-point walk of single_grid_search: the reference's solver


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def close(a, b, tol):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    ok = ~np.isnan(b)
    return (not ok.any()) or float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(b[ok])))) <= tol


t0 = time.time()
bad = 0
for i in range(cases):
    # ---- grid_search, multiple-graph problems (grids of at least 2 x 2: the reference squeezes its meshgrid) ----
    p, K = int(rng.choice([5, 8, 12, 17])), int(rng.integers(2, 4))
    reg = "GGL" if rng.random() < 0.5 else "FGL"
    n1, n2 = int(rng.integers(2, 4)), int(rng.integers(2, 4))
    l1 = np.sort(10.0 ** rng.uniform(-1.5, -0.3, n1))[::-1].copy()
    use_w2 = bool(rng.random() < 0.5)
    l2 = None if use_w2 else np.sort(10.0 ** rng.uniform(-2.0, -0.5, n2))[::-1].copy()
    w2 = np.sort(rng.uniform(0.05, 0.6, n2)) if use_w2 else None
    latent, thr = bool(rng.random() < 0.4), bool(rng.random() < 0.4)
    method, gamma = ("eBIC" if rng.random() < 0.6 else "AIC"), float(rng.choice([0.1, 0.3, 0.5]))
    Nk = rng.integers(p, 4 * p + 5, K)
    S, _ = synth.make_problem(reg, K, p, N=int(Nk.max()), seed=int(rng.integers(1 << 30)))
    kw = dict(method=method, gamma=gamma, latent=latent, thresholding=thr, tol=1e-7, rtol=1e-7)
    if latent:
        nmu = int(rng.integers(1, 4))
        kw["mu_range"] = np.sort(10.0 ** rng.uniform(-0.3, 0.7, nmu))[::-1].copy()
        kw["ix_mu"] = rng.integers(0, nmu, (K, n1))
    tag = dict(i=i, p=p, K=K, reg=reg, n1=n1, n2=n2, w2=use_w2, latent=latent, thr=thr, method=method, gamma=gamma)
    rst, rix, rbest = quiet(rms.grid_search, admm.ADMM_MGL, S, Nk, p, reg, l1, l2=l2, w2=w2, **kw)
    st, ix, best = quiet(ms.grid_search, admm.ADMM_MGL, S, Nk, p, reg, l1, l2=l2, w2=w2, batched=False, **kw)
    probs = []
    for nm in ("AIC", "SP") + (("RANK",) if latent else ()) + (("TAU",) if thr else ()):
        if nm in rst and rst[nm] is not None and not close(st[nm], rst[nm], 1e-10):
            probs.append(nm)
    # (the reference APPENDS every call's gamma to its module-level default list, :161-163: its tables accumulate the gammas of
    # earlier calls -- compared are the ones this call asked for)
    assert set(st["BIC"]) <= set(rst["BIC"])
    for g in st["BIC"]:
        if not close(st["BIC"][g], rst["BIC"][g], 1e-10):
            probs.append(f"BIC[{g}]")
    if tuple(np.atleast_1d(ix).tolist()) != tuple(np.atleast_1d(rix).tolist()):
        probs.append(f"ix {ix} vs {rix}")
    if not close(best["Theta"], rbest["Theta"], 1e-10):
        probs.append("best Theta")
    if probs:
        bad += 1
        print("grid_search OFF", probs, tag, flush=True)

    # ---- single_grid_search, point by point (a mask of ones takes that branch here; use_block=False there: ADMM_SGL on both sides) ----
    p = int(rng.choice([4, 7, 12, 20]))
    nl = int(rng.integers(1, 5))
    latent, thr = bool(rng.random() < 0.5), bool(rng.random() < 0.4)
    nm_ = int(rng.integers(1, 4)) if latent else 1
    lam = np.sort(10.0 ** rng.uniform(-1.8, -0.2, nl))[::-1].copy()
    mu = np.sort(10.0 ** rng.uniform(-0.3, 0.8, nm_))[::-1].copy() if latent else None
    method, gamma = ("eBIC" if rng.random() < 0.6 else "AIC"), float(rng.choice([0.1, 0.3, 0.5, 0.25]))
    Ns = int(rng.integers(p, 4 * p + 5))
    S1, _ = synth.make_problem("GGL", 1, p, N=Ns, seed=int(rng.integers(1 << 30)))
    mask = np.ones((p, p))
    if rng.random() < 0.4:
        M = rng.random((p, p)) < 0.7
        mask = ((M | M.T) * 1.0 + 0.5)
    tag = dict(i=i, p=p, nl=nl, nm=nm_, latent=latent, thr=thr, method=method, gamma=gamma, ones=bool(np.all(mask == 1)))
    kw = dict(method=method, gamma=gamma, latent=latent, mu_range=mu, thresholding=thr, tol=1e-8, rtol=1e-8, lambda1_mask=mask)
    rb, re, rl, rs = quiet(rms.single_grid_search, S1[0], lam, Ns, use_block=False, **kw)
    b, e, l, s = quiet(ms.single_grid_search, S1[0], lam, Ns, **kw)
    probs = [nm for nm in ("AIC", "SP", "RANK") if not close(s[nm], rs[nm], 1e-10)]
    assert set(s["BIC"]) <= set(rs["BIC"])
    probs += [f"BIC[{g}]" for g in s["BIC"] if not close(s["BIC"][g], rs["BIC"][g], 1e-10)]
    if thr and "TAU" in rs and rs["TAU"] is not None and not close(s["TAU"], rs["TAU"], 1e-12):
        probs.append("TAU")
    if [float(s["BEST"]["lambda1"]), float(s["BEST"]["mu1"])] != [float(rs["BEST"]["lambda1"]), float(rs["BEST"]["mu1"])]:
        probs.append(f"BEST {s['BEST']} vs {rs['BEST']}")
    if not close(e, re, 1e-10) or (latent and not close(l, rl, 1e-10)) or not close(b["Theta"], rb["Theta"], 1e-10):
        probs.append("estimates / lowrank / best Theta")
    if probs:
        bad += 1
        print("single_grid_search OFF", probs, tag, flush=True)
print(f"{cases} random grids each for grid_search and single_grid_search (seed {seed}), our host walk vs the reference's, the reference's solver on both "
      f"sides: {bad} differ; {time.time() - t0:.0f} s")
print("ok" if bad == 0 else f"{bad} findings")
