"""Randomised parity cases shared by tests/test_gpu_fuzz.py and tools/fuzz_parity.py: ADMM_MGL / ADMM_SGL through the C ABI against
the CPU oracle (test infrastructure) on shapes and parameters the fixed tests do not visit -- p in 1 .. 130 (one case in ten
191 .. 300) around every tile / pair / wave boundary, K in 1 .. 9, both penalties, latent on / off, penalty masks, N < p (singular
S), lambda from "nothing shrinks" to "everything does", rho over three decades, rho updates on / off, fixed-length runs at
tol = 1e-20 and runs to a realistic tolerance.
Reference behaviour: solver/admm_solver.py:13-313 (ADMM_MGL), solver/single_admm_solver.py:15-275 (ADMM_SGL)."""
import contextlib
import io
import os
import time
import warnings

import numpy as np

P = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 23, 31, 32, 33, 47, 63, 64, 65, 66, 95, 97, 127, 128, 129, 130]
PBIG = [191, 200, 255, 256, 257, 300]          # (one case in ten)
TOL = float(os.environ.get("GGL_FUZZ_TOL", 1e-9))    # relative to max(1, |reference|_max), every array of the solution (the
                                               # variable: a tighter bar lists the cases nearest to it)
LAST = {}                                      # the inputs of the case under way: written out when it is off
N_SAMPLES_INT = False                          # see one(): True in tests/golden/fuzz_oracle_vs_reference.py
GROUPED = []                                   # one_bigbatch: (grouping forced, Omega-steps that ran as groups) per case


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn(*a, **k)


def one(i, rng):
    from gglasso_amd import solver, synth
    from oracle import ggl_oracle as orc
    p = int(rng.choice(PBIG)) if rng.random() < 0.1 else int(rng.choice(P))
    single = rng.random() < 0.3
    reg = "GGL" if rng.random() < 0.5 else "FGL"
    K = 1 if single else int(rng.integers(2, 10))
    latent = rng.random() < 0.4
    lam1 = float(10.0 ** rng.uniform(-3, 0.3))
    lam2 = float(10.0 ** rng.uniform(-3, 0.0))
    mu1 = float(10.0 ** rng.uniform(-1, 1))
    rho = float(10.0 ** rng.uniform(-1.5, 1.5))
    upd = bool(rng.random() < 0.5)
    iters = int(rng.integers(1, 25))
    conv = bool(rng.random() < 0.3)                           # a run to a realistic tolerance instead of a fixed length
    tol, rtol = (float(10.0 ** rng.uniform(-9, -5)), float(10.0 ** rng.uniform(-8, -4))) if conv else (1e-20, 1e-20)
    if conv:
        iters = 400
    S, _ = synth.make_problem(reg, K, p, N=int(rng.integers(max(2, p // 2), 3 * p + 3)), seed=int(rng.integers(1 << 30)))
    # (N < p: a singular S, as the reference meets it in high-dimensional use)
    tag = dict(i=i, p=p, K=K, reg=reg if not single else "SGL", latent=latent, lam1=round(lam1, 5), lam2=round(lam2, 5),
               mu1=round(mu1, 4), rho=round(rho, 4), upd=upd, iters=iters, tol=tol, rtol=rtol)
    kw = dict(max_iter=iters, tol=tol, rtol=rtol, rho=rho, update_rho=upd, latent=latent, measure=True)
    # a third of the cases start from somewhere else: Omega_0 positive definite, Theta_0 another matrix, X_0 a symmetric dual
    warm = bool(rng.random() < 0.33)
    starts = None
    if warm:
        def spd():
            A = rng.standard_normal((K, p, p + 2))
            return A @ A.transpose(0, 2, 1) / (p + 2) + 0.1 * np.eye(p)
        X0 = rng.standard_normal((K, p, p)) * 0.1
        starts = (spd(), spd(), 0.5 * (X0 + X0.transpose(0, 2, 1)))
        tag["warm"] = True
    # (the KKT criterion: eta_A <= tol decides, admm_solver.py:197-204 / single_admm_solver.py:199-206)
    if conv and rng.random() < 0.3:
        kw["stopping_criterion"] = "kkt"
        kw["tol"] = tol = float(10.0 ** rng.uniform(-6, -3))
        tag.update(kkt=True, tol=tol)
    LAST.clear()
    LAST.update(S=S, lam1=lam1, lam2=lam2, single=single, reg=reg, **{k: v for k, v in kw.items() if k != "measure"})
    if latent:
        kw["mu1"] = mu1
    if single:
        eye = np.eye(p)
        if rng.random() < 0.3:                                  # an entrywise penalty mask (single_admm_solver.py:68-73)
            M = rng.random((p, p)) < 0.6
            kw["lambda1_mask"] = ((M | M.T) * rng.uniform(0.5, 2.0)).astype(float)
            tag["mask"] = True
        if warm:
            eye, kw["Theta_0"], kw["X_0"] = starts[0][0], starts[1][0], starts[2][0]
        ref, rinfo = quiet(orc.ADMM_SGL, S[0], lam1, eye.copy(), **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
        got, ginfo = quiet(solver.ADMM_SGL, S[0], lam1, eye.copy(), **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    else:
        eye = np.repeat(np.eye(p)[None], K, axis=0)
        if warm:
            eye, kw["Theta_0"], kw["X_0"] = starts
        if rng.random() < 0.3:
            # the weights of the log-likelihood terms (admm_solver.py:127-139): an array (K,) -- or one int for all where the
            # REFERENCE itself is on the other side (its array branch cannot be reached: `n_samples == None` on an array raises)
            kw["n_samples"] = int(rng.integers(5, 500)) if N_SAMPLES_INT else rng.integers(5, 500, K)
            tag["n_samples"] = True
        ref, rinfo = quiet(orc.ADMM_MGL, S, lam1, lam2, reg, eye.copy(), **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
        got, ginfo = quiet(solver.ADMM_MGL, S, lam1, lam2, reg, eye.copy(), **{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    worst = 0.0
    # past convergence with tol = 1e-20 the residuals are rounding noise and so are the rho updates they decide (r > 10 s or
    # s > 10 r, admm_solver.py:186-195): the two runs then scale the dual differently -- X = (unscaled dual) / rho is compared
    # only while the residuals still mean something
    last = np.abs(np.asarray(rinfo["residual"][-1], dtype=float))
    noise = (not conv) and upd and float(np.max(last)) < 1e-11 * max(1.0, float(np.abs(np.asarray(ref["Theta"])).max()))
    if noise:
        tag["rho_by_noise"] = True
    for nm in ("Omega", "Theta") + (() if noise else ("X",)) + (("L",) if latent else ()):
        a, b = np.asarray(got[nm]), np.asarray(ref[nm])
        if a.shape != b.shape:
            return tag, float("inf"), f"shape {nm} {a.shape} vs {b.shape}"
        if not np.all(np.isfinite(a)):
            return tag, float("inf"), f"non-finite {nm}"
        worst = max(worst, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
    note = ""
    nit_g, nit_r = len(ginfo["residual"]), len(rinfo["residual"])
    if conv and nit_g != nit_r:
        # a stopping test decided by the last bits of a residual: one iteration apart is the same solve, the iterates then
        # differ by about the tolerance
        return tag, (0.0 if abs(nit_g - nit_r) <= 1 else float("inf")), f"iterations {nit_g} vs {nit_r}"
    if ginfo["status"] != rinfo["status"]:
        # with tol = rtol = 1e-20 only a residual that is EXACTLY zero in one of the two passes its test: a last-bit matter
        exact0 = not conv
        note = f"status {ginfo['status']!r} vs {rinfo['status']!r}" + (" (exact-zero residual)" if exact0 else "")
    return tag, worst, note



def one_batch(i, rng):
    """A grid of G points through ADMM_SGL_batch / ADMM_MGL_batch (per-point rho, stopping decision, snapshot at the iteration a
    point converges, compaction of the stack) against the oracle's independent solve of every point (what the reference's grid
    walks do one point after the other, helper/model_selection.py:208-224, 619-633)."""
    from gglasso_amd import batch, synth
    from oracle import ggl_oracle as orc
    p = int(rng.choice([q for q in P if q >= 2]))
    single = rng.random() < 0.5
    reg = "GGL" if rng.random() < 0.5 else "FGL"
    K = 1 if single else int(rng.integers(2, 6))
    G = int(rng.integers(1, 13))
    latent = rng.random() < 0.4
    lam1 = 10.0 ** rng.uniform(-2.5, 0.0, G)
    lam2 = 10.0 ** rng.uniform(-2.5, -0.3, G)
    mu = 10.0 ** rng.uniform(-0.7, 0.7, G)
    upd = bool(rng.random() < 0.7)
    tol, rtol = float(10.0 ** rng.uniform(-9, -6)), float(10.0 ** rng.uniform(-8, -5))
    max_iter = int(rng.choice([15, 60, 400]))               # (15: most points end at the iteration limit)
    compact = bool(rng.random() < 0.7)
    S, _ = synth.make_problem(reg, K, p, N=int(rng.integers(max(2, p // 2), 3 * p + 3)), seed=int(rng.integers(1 << 30)))
    tag = dict(i=i, kind="batch", p=p, K=K, G=G, reg=reg if not single else "SGL", latent=latent, upd=upd, tol=tol, rtol=rtol,
               max_iter=max_iter, compact=compact)
    LAST.clear()
    LAST.update(S=S, lam1=lam1, lam2=lam2, mu=mu, single=single, reg=reg, latent=latent, upd=upd, tol=tol, rtol=rtol,
                max_iter=max_iter, compact=compact)
    kw = dict(tol=tol, rtol=rtol, update_rho=upd, max_iter=max_iter, latent=latent)
    if single:
        eye = np.eye(p)
        res = quiet(batch.ADMM_SGL_batch, S[0], lam1, Omega_0=eye, X_0=np.zeros((p, p)), mu1=mu if latent else None,
                    compact=compact, **kw)
    else:
        eye = np.repeat(np.eye(p)[None], K, axis=0)
        res = quiet(batch.ADMM_MGL_batch, S, lam1, lam2, reg, Omega_0=eye, mu1=np.repeat(mu[:, None], K, axis=1) if latent else None,
                    compact=compact, **kw)
    worst, note = 0.0, ""
    for g in range(G):
        okw = dict(kw)
        if latent:
            okw["mu1"] = float(mu[g]) if single else np.full(K, mu[g])
        if single:
            ref, rinfo = quiet(orc.ADMM_SGL, S[0], float(lam1[g]), eye, X_0=np.zeros((p, p)), **okw)
        else:
            ref, rinfo = quiet(orc.ADMM_MGL, S, float(lam1[g]), float(lam2[g]), reg, eye, **okw)
        sol, info = res[g]
        if info["iterations"] != rinfo["iterations"]:
            # a stopping test decided by the last bits of a residual: one iteration apart is the same solve
            if abs(info["iterations"] - rinfo["iterations"]) > 1:
                return tag, float("inf"), f"point {g}: iterations {info['iterations']} vs {rinfo['iterations']}"
            note = f"point {g}: iterations {info['iterations']} vs {rinfo['iterations']}"
            continue
        if info["status"] != rinfo["status"]:
            return tag, float("inf"), f"point {g}: status {info['status']!r} vs {rinfo['status']!r}"
        for nm in ("Omega", "Theta", "X") + (("L",) if latent else ()):
            a, b = np.asarray(sol[nm]), np.asarray(ref[nm])
            if a.shape != b.shape or not np.all(np.isfinite(a)):
                return tag, float("inf"), f"point {g}: {nm} shape {a.shape} vs {b.shape} or not finite"
            worst = max(worst, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
    return tag, worst, note


def one_block(i, rng):
    """block_SGL (solver/single_admm_solver.py:326-475): a covariance matrix with planted components of random sizes (singletons,
    pairs, blocks of very different size in one problem), lambda1 between the planted cross-block level and the in-block
    correlations, optional mask; all components solved together on the GPU against the oracle's component-by-component loop."""
    from gglasso_amd import solver
    from oracle import ggl_oracle as orc
    nb = int(rng.integers(1, 7))
    sizes = [int(rng.choice([1, 1, 2, 3, 5, 8, 13, 21, 40, 70])) for _ in range(nb)]
    p = int(sum(sizes))
    blocks = []
    for n in sizes:
        A = rng.standard_normal((n, 3 * n + 2))
        blocks.append(A @ A.T / (3 * n + 2))
    from scipy.linalg import block_diag
    S = block_diag(*blocks)
    noise = rng.standard_normal((p, p)) * 1e-3
    S = S + (noise + noise.T) / 2 * (S == 0)                     # cross-block entries far below any lambda1 drawn here
    perm = rng.permutation(p)
    S = np.ascontiguousarray(S[np.ix_(perm, perm)])
    lam1 = float(10.0 ** rng.uniform(-1.7, -0.3))
    tol, rtol = float(10.0 ** rng.uniform(-9, -6)), float(10.0 ** rng.uniform(-8, -4))
    upd = bool(rng.random() < 0.7)
    rho = float(10.0 ** rng.uniform(-0.5, 0.5))
    max_iter = int(rng.choice([20, 500]))
    kw = dict(tol=tol, rtol=rtol, update_rho=upd, rho=rho, max_iter=max_iter)
    tag = dict(i=i, kind="block", p=p, sizes=sizes, lam1=round(lam1, 5), **kw)
    if rng.random() < 0.3:
        M = rng.random((p, p)) < 0.7
        kw["lambda1_mask"] = ((M | M.T) * 1.0 + 0.5).astype(float)
        tag["mask"] = True
    LAST.clear()
    LAST.update(S=S, lam1=lam1, **kw)
    ref = quiet(orc.block_SGL, S, lam1, np.eye(p), **kw)
    got = quiet(solver.block_SGL, S, lam1, np.eye(p), **kw)
    worst = 0.0
    for nm in ("Omega", "Theta", "X"):
        a, b = np.asarray(got[nm]), np.asarray(ref[nm])
        if a.shape != b.shape or not np.all(np.isfinite(a)):
            return tag, float("inf"), f"{nm}: shape {a.shape} vs {b.shape} or not finite"
        worst = max(worst, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
    # (a component whose stopping test is decided by the last bits ends one iteration apart: about the tolerance, not 1e-9)
    note = ""
    if worst > TOL and worst <= 50 * max(tol, rtol):
        note, worst = f"iterations: one component differs by {worst:.1e} <= 50 max(tol, rtol)", 0.0
    return tag, worst, note


def _nonconforming(rng, universe, sizes, prob=0.08):
    """Instances observing random subsets of a common variable set; G like helper/ext_admm_helper.py:104-144 (one group per
    pair of variables present together in >= 2 instances)."""
    members = [np.sort(rng.choice(universe, size=n, replace=False)) for n in sizes]
    A = rng.standard_normal((universe, universe)) * (rng.random((universe, universe)) < prob)
    Sig = np.linalg.inv(A @ A.T + np.eye(universe))
    S = {}
    for k, ix in enumerate(members):
        X = rng.multivariate_normal(np.zeros(universe), Sig, size=3 * len(ix) + 2).T[ix]
        S[k] = np.atleast_2d(np.cov(X, bias=True))
    K = len(sizes)
    loc = -np.ones((universe, K), dtype=int)
    for k, ix in enumerate(members):
        loc[ix, k] = np.arange(len(ix))
    rows = []
    for a in range(universe):
        for b in range(a + 1, universe):
            both = (loc[a] >= 0) & (loc[b] >= 0)
            if both.sum() >= 2:
                rows.append((np.where(both, np.minimum(loc[a], loc[b]), -1), np.where(both, np.maximum(loc[a], loc[b]), -1)))
    if not rows:
        return None
    G = np.stack([np.stack([r[0] for r in rows]), np.stack([r[1] for r in rows])]).astype(int)
    return S, G, np.array([len(ix) for ix in members])


def one_ext(i, rng):
    """ext_ADMM_MGL (solver/ext_admm_solver.py:18-323): K = 2 .. 5 instances of DIFFERENT dimension over a common variable set,
    random group structure, per-instance lambda1 / mu1, latent on / off, fixed lengths and runs to a tolerance."""
    from gglasso_amd.ext_solver import ext_ADMM_MGL
    from oracle import ggl_oracle as orc
    K = int(rng.integers(2, 6))
    universe = int(rng.choice([6, 9, 14, 20, 33, 48, 70]))
    sizes = [int(rng.integers(max(2, universe // 3), universe + 1)) for _ in range(K)]
    made = _nonconforming(rng, universe, sizes, prob=float(rng.uniform(0.05, 0.3)))
    tag = dict(i=i, kind="ext", K=K, universe=universe, sizes=sizes)
    if made is None:
        return tag, 0.0, ""
    S, G, p = made
    orc.check_G(G, p)
    latent = rng.random() < 0.4
    lam1 = 10.0 ** rng.uniform(-2.0, -0.5, K)
    lam2 = float(10.0 ** rng.uniform(-2.5, -0.5))
    rho = float(10.0 ** rng.uniform(-0.7, 0.7))
    conv = bool(rng.random() < 0.3)
    tol, rtol = (float(10.0 ** rng.uniform(-8, -5)), float(10.0 ** rng.uniform(-7, -4))) if conv else (1e-20, 1e-20)
    iters = 300 if conv else int(rng.integers(1, 16))
    kw = dict(max_iter=iters, tol=tol, rtol=rtol, rho=rho, latent=latent, measure=True)
    if latent:
        kw["mu1"] = 10.0 ** rng.uniform(-0.7, 0.5, K)
    tag.update(latent=latent, lam2=round(lam2, 5), rho=round(rho, 4), iters=iters, tol=tol, rtol=rtol, groups=int(G.shape[1]))
    LAST.clear()
    LAST.update(G=G, lam1=lam1, lam2=lam2, **{f"S_{k}": S[k] for k in range(K)}, **{k: v for k, v in kw.items() if k != "measure"})
    Om0 = {k: np.eye(p[k]) for k in range(K)}
    ref, rinfo = quiet(orc.ext_ADMM_MGL, S, lam1, lam2, 'GGL', {k: v.copy() for k, v in Om0.items()}, G, **kw)
    got, ginfo = quiet(ext_ADMM_MGL, S, lam1, lam2, 'GGL', {k: v.copy() for k, v in Om0.items()}, G, **kw)
    nit_g, nit_r = len(ginfo["residual"]), len(rinfo["residual"])
    if nit_g != nit_r:
        return tag, (0.0 if conv and abs(nit_g - nit_r) <= 1 else float("inf")), f"iterations {nit_g} vs {nit_r}"
    note = ""
    if ginfo["status"] != rinfo["status"]:
        note = f"status {ginfo['status']!r} vs {rinfo['status']!r}" + ("" if conv else " (exact-zero residual)")
    worst = 0.0
    for nm in ("Omega", "Theta", "X0", "X1") + (("L",) if latent else ()):
        for k in range(K):
            a, b = np.asarray(got[nm][k]), np.asarray(ref[nm][k])
            if a.shape != b.shape or not np.all(np.isfinite(a)):
                return tag, float("inf"), f"{nm}[{k}]: shape {a.shape} vs {b.shape} or not finite"
            worst = max(worst, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
    return tag, worst, note


def _spectrum(rng, p):
    """Eigenvalues with what a random matrix never has: wide ranges, clusters, exact repeats, zeros, one sign only."""
    kind = int(rng.integers(0, 6))
    if kind == 0:
        d = rng.standard_normal(p) * 10.0 ** rng.uniform(-2, 2)
    elif kind == 1:
        d = 10.0 ** rng.uniform(-4, 3, p) * rng.choice([-1.0, 1.0], p)           # seven decades, both signs
    elif kind == 2:
        d = np.repeat(rng.standard_normal(max(1, p // 8 + 1)), 8)[:p] * 10.0 ** rng.uniform(-1, 1)   # exact repeats
    elif kind == 3:
        d = 1.0 + 1e-9 * rng.standard_normal(p)                                # one cluster
    elif kind == 4:
        d = np.abs(rng.standard_normal(p)) * (rng.random(p) < 0.5)             # semidefinite, many exact zeros
    else:
        d = -np.abs(rng.standard_normal(p)) * 10.0 ** rng.uniform(-1, 2)       # negative definite
    return d


def one_ops(i, rng):
    """The operators on their own (solver/ggl_helper.py:16-36, 190-207, 280-303; solver/fgl_helper.py:11-68) at inputs ADMM iterates
    rarely are: phiplus and prox_rank_norm from the MATRIX (the Omega- / L-step the solver runs, method by the size rule) on
    stacks whose instances have engineered spectra of very different conditioning, prox_p with ties and exact zeros across K."""
    from gglasso_amd import ops
    from oracle import ggl_oracle as orc
    which = int(rng.integers(0, 3))
    K = int(rng.integers(1, 9))
    tag = dict(i=i, kind="ops", K=K)
    if which < 2:
        p = int(rng.choice([q for q in P + [200, 257, 300] if q >= 2]))
        W = np.empty((K, p, p))
        for k in range(K):
            Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
            W[k] = (Q * _spectrum(rng, p)) @ Q.T
            W[k] = 0.5 * (W[k] + W[k].T)
        beta = 10.0 ** rng.uniform(-3, 1.5, K)
        tag.update(op="phiplus" if which == 0 else "rank", p=p, beta=[float(f"{b:.3g}") for b in beta])
        LAST.clear()
        LAST.update(W=W, beta=beta, which=which)
        if which == 0:
            ref, _ = orc.phiplus_stack(W, beta)
            out = ops.phiplus_matrix(W, beta)
        else:
            ref = orc.rank_stack(W, beta)
            out = ops.rank_matrix(W, beta)
        if not np.all(np.isfinite(out)):
            return tag, float("inf"), "not finite"
        # per instance: relative to the larger of 1, the result and the input (an eigenvalue-wise map: its error scales with |W|)
        worst = max(float(np.abs(out[k] - ref[k]).max()) / max(1.0, float(np.abs(ref[k]).max()), float(np.abs(W[k]).max()))
                    for k in range(K))
        if not np.array_equal(out, out.transpose(0, 2, 1)):
            return tag, float("inf"), "result not bitwise symmetric"
        return tag, worst, ""
    p = int(rng.choice(P))
    reg = "GGL" if rng.random() < 0.5 else "FGL"
    K = max(K, 2)
    X = rng.standard_normal((K, p, p))
    style = int(rng.integers(0, 3))
    if style == 1:
        X = np.round(X * 2) / 2                                    # ties across K and exact zeros
    elif style == 2:
        X = np.repeat(X[:1], K, axis=0) + (rng.random((K, 1, 1)) < 0.3) * 1e-14   # (almost) constant across K
    X = 0.5 * (X + X.transpose(0, 2, 1))
    l1, l2 = float(10.0 ** rng.uniform(-3, 0.5)), float(10.0 ** rng.uniform(-3, 0.5))
    tag.update(op="prox_p", reg=reg, p=p, K=K, style=style, l1=l1, l2=l2)
    LAST.clear()
    LAST.update(X=X, l1=l1, l2=l2, reg=reg)
    ref = orc.prox_p(X, l1, l2, reg)
    out = ops.prox_p(X, l1, l2, reg)
    return tag, float(np.abs(out - ref).max()) / max(1.0, float(np.abs(ref).max())), ""


def one_stats(i, rng):
    """What the AIC / eBIC / rank tables of model selection are made of (helper/model_selection.py:619-660, 698-737, 884-894):
    <S,Theta>, log det Theta, count_nonzero(Theta), lambda_min(Theta), matrix_rank(L) and the same statistics of the estimate
    thresholded at every tau, computed on the device beside a random grid -- against numpy on the solutions the grid returns."""
    from gglasso_amd import batch, synth
    p = int(rng.choice([q for q in P if q >= 3]))
    G = int(rng.integers(1, 10))
    latent = rng.random() < 0.5
    lam = 10.0 ** rng.uniform(-2.0, 0.0, G)
    mu = 10.0 ** rng.uniform(-0.5, 0.7, G)
    with_tau = rng.random() < 0.5
    tau = np.sort(10.0 ** rng.uniform(-8, -0.5, int(rng.integers(1, 8)))) if with_tau else None
    tol = float(10.0 ** rng.uniform(-9, -6))
    S, _ = synth.make_problem("GGL", 1, p, N=int(rng.integers(p, 3 * p + 3)), seed=int(rng.integers(1 << 30)))
    S = S[0]
    tag = dict(i=i, kind="stats", p=p, G=G, latent=latent, ntau=0 if tau is None else len(tau), tol=tol)
    LAST.clear()
    LAST.update(S=S, lam=lam, mu=mu, latent=latent, tau=tau, tol=tol)
    res = quiet(batch.ADMM_SGL_batch, S, lam, Omega_0=np.eye(p), X_0=np.zeros((p, p)), tol=tol, rtol=tol, max_iter=500,
                latent=latent, mu1=mu if latent else None, selection_stats=True, tau_range=tau)
    worst = 0.0

    def stats_of(T):
        d = np.linalg.eigvalsh(T)
        return float(np.sum(S * T)), float(np.linalg.slogdet(T)[1]), int(np.count_nonzero(T)), float(d.min())

    def cmp(got4, T, what):
        nonlocal worst
        sd, ld, nz, lm = stats_of(T)
        if int(got4[2]) != nz:
            return f"{what}: nnz {int(got4[2])} vs {nz}"
        worst = max(worst, abs(got4[0] - sd) / max(1.0, abs(sd)), abs(got4[3] - lm) / max(1.0, np.abs(T).max()))
        if lm > 1e-9:                                  # (log det of a matrix that is not definite is -inf / nan by convention)
            if not np.isfinite(got4[1]):
                return f"{what}: log det {got4[1]} for a definite matrix (lambda_min {lm:.3e})"
            worst = max(worst, abs(got4[1] - ld) / max(1.0, abs(ld)))
        return ""

    for g, (sol, info) in enumerate(res):
        if info["status"] == "solver error":
            return tag, float("inf"), f"point {g}: solver error {info.get('error')}"
        sel = info["selection"]
        bad = cmp([sel["Sdot"], sel["logdet"], sel["nnz"], sel["lambda_min"]], sol["Theta"], f"point {g}")
        if bad:
            return tag, float("inf"), bad
        if latent:
            sv = np.linalg.svd(sol["L"], compute_uv=False)
            cut = sv.max() * p * np.finfo(float).eps if sv.size else 0.0
            clear = not np.any((sv > cut / 30) & (sv < cut * 30))       # (a singular value AT numpy's cut is anybody's call)
            if clear and int(sel["rank"]) != int(np.linalg.matrix_rank(sol["L"])):
                return tag, float("inf"), f"point {g}: rank {sel['rank']} vs {np.linalg.matrix_rank(sol['L'])}"
        if tau is not None:
            tab = np.asarray(sel["threshold"])
            for t in range(len(tau)):
                Tt = sol["Theta"] * ((np.abs(sol["Theta"]) > tau[t]) | np.eye(p, dtype=bool))
                bad = cmp(tab[t], Tt, f"point {g}, tau {tau[t]:.3g}")
                if bad:
                    return tag, float("inf"), bad
    return tag, worst, ""


def _pattern_note(aic, raic, what="zero pattern"):
    """Two tables whose zero patterns differ: a last-bit matter only if the fit terms agree, i.e. the AIC tables differ by
    whole edges (an entry whose prox input sits AT the threshold -- a weakly active entry, e.g. of a group the group
    penalty has zeroed in ext_ADMM_MGL, where ADMM's dual ends on the boundary of its feasible set -- is +-1 ulp in one
    evaluation and exactly zero in another: 2^-58 .. 2^-56 seen).  Anything else is a finding."""
    ok = ~(np.isnan(aic) | np.isnan(raic))
    d = 2.0 * (aic[ok] - raic[ok])          # in half edges: the oracle's Theta (like the reference's) is not bitwise symmetric -- its
    # Omega is Q diag Q^T from a matrix product -- so ONE of a pair (i,j), (j,i) can pass the threshold by an ulp; ours is symmetric
    if np.all(np.abs(d - np.round(d)) <= 2e-5 * np.maximum(1.0, np.abs(raic[ok]))):
        return 0.0, f"iterations: {what} differs by entries at the threshold (last bit): {int(np.abs(np.round(d)).sum())} entries in the table"
    return float("inf"), f"{what} differs and the AIC tables are not whole entries apart (max fractional part {float(np.max(np.abs(d - np.round(d)))) / 2:.2e})"


def one_grid(i, rng):
    """single_grid_search (helper/model_selection.py:505-692) on a random (lambda1, mu1) grid -- all points one batch, tables from
    the device statistics -- against the same tables built on the host (numpy formulas of _grid_tables' point-by-point branch)
    from the oracle's independent solve of every point from the reference's start Omega_0 = X_0 = I (:595-596)."""
    from gglasso_amd import model_selection as ms, synth
    from oracle import ggl_oracle as orc
    p = int(rng.choice([q for q in P if 3 <= q <= 97]))
    nl, latent = int(rng.integers(1, 6)), bool(rng.random() < 0.5)
    nm = int(rng.integers(1, 4)) if latent else 1
    lam = np.sort(10.0 ** rng.uniform(-1.8, -0.2, nl))[::-1].copy()
    mu = np.sort(10.0 ** rng.uniform(-0.3, 0.8, nm))[::-1].copy() if latent else None
    method = "eBIC" if rng.random() < 0.6 else "AIC"
    gamma = float(rng.choice([0.1, 0.3, 0.5, 0.7, 0.25]))
    thr = bool(rng.random() < 0.4)
    tol = float(10.0 ** rng.uniform(-9, -7))
    Ns = int(rng.integers(p, 4 * p + 5))
    S, _ = synth.make_problem("GGL", 1, p, N=Ns, seed=int(rng.integers(1 << 30)))
    S = S[0]
    tag = dict(i=i, kind="grid", p=p, nl=nl, nm=nm, latent=latent, method=method, gamma=gamma, thresholding=thr, tol=tol, N=Ns)
    LAST.clear()
    LAST.update(S=S, lam=lam, mu=mu, latent=latent, method=method, gamma=gamma, thr=thr, tol=tol, N=Ns)
    best, est, low, st = quiet(ms.single_grid_search, S, lam, Ns, method=method, gamma=gamma, latent=latent, mu_range=mu,
                               thresholding=thr, tol=tol, rtol=tol)
    sols = []
    eye = np.eye(p)
    for j in range(nl):
        for m in range(nm):
            kw = dict(latent=True, mu1=float(mu[m])) if latent else {}
            sol, _ = quiet(orc.ADMM_SGL, S, float(lam[j]), eye, X_0=eye, tol=tol, rtol=tol, max_iter=1000, **kw)
            sols.append(sol)
    mu_r = mu if latent else np.array([0])
    gammas = sorted(set(ms.DEFAULT_GAMMAS) | {gamma})
    rbest, rest, rlow, rst = ms._grid_tables(S, Ns, sols, None, lam, mu_r, latent, method, gamma, gammas, True, None, thr)
    loose = max(TOL, 100 * tol)                       # (a point that stops one iteration apart is off by about its tolerance)
    d_est = float(np.abs(est - rest).max()) / max(1.0, float(np.abs(rest).max()))
    if d_est > loose:
        return tag, float("inf"), f"estimates differ by {d_est:.2e}"
    if latent:
        d_low = float(np.abs(low - rlow).max()) / max(1.0, float(np.abs(rest).max()))
        if d_low > loose:
            return tag, float("inf"), f"lowrank differs by {d_low:.2e}"
    if thr and not np.array_equal(st["TAU"], rst["TAU"]):
        return tag, 0.0, "iterations: another threshold chosen at a point (scores of two candidates within the solve's tolerance)"
    if not np.array_equal(st["SP"], rst["SP"]):
        return (tag,) + _pattern_note(st["AIC"], rst["AIC"])
    worst = 0.0
    for a, b, nm_ in [(st["AIC"], rst["AIC"], "AIC")] + [(st["BIC"][g], rst["BIC"][g], f"BIC[{g}]") for g in gammas]:
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            return tag, float("inf"), f"{nm_}: NaN pattern differs"
        ok = ~np.isnan(b)
        if ok.any():
            # (the fit term N (<S,Theta> - log det Theta) inherits the solve's tolerance times N p)
            worst_t = float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(b[ok]))))
            if worst_t > max(1e-7, 1e3 * tol):
                return tag, float("inf"), f"{nm_} differs by {worst_t:.2e}"
    if latent and not np.array_equal(st["RANK"], rst["RANK"]):
        # numpy's rule on a singular value at its cut is anybody's call; anything else is a finding
        for j in range(nl):
            for m in range(nm):
                if st["RANK"][j, m] != rst["RANK"][j, m]:
                    sv = np.linalg.svd(rlow[j, m], compute_uv=False)
                    cut = sv.max() * p * np.finfo(float).eps
                    if not np.any((sv > cut / 1e3) & (sv < cut * 1e3)):
                        return tag, float("inf"), f"RANK[{j},{m}] {st['RANK'][j, m]} vs {rst['RANK'][j, m]}"
        return tag, 0.0, "iterations: rank differs at a singular value within 1e3 of numpy's cut"
    if [float(st["BEST"]["lambda1"]), float(st["BEST"]["mu1"])] != [float(rst["BEST"]["lambda1"]), float(rst["BEST"]["mu1"])]:
        tab = rst["AIC"] if method == "AIC" else rst["BIC"][gamma]
        two = np.sort(tab[~np.isnan(tab)])[:2]
        if len(two) == 2 and abs(two[1] - two[0]) <= 1e-6 * max(1.0, abs(two[0])):
            return tag, 0.0, "iterations: best point differs between two scores that tie"
        return tag, float("inf"), f"BEST {st['BEST']} vs {rst['BEST']}"
    d_best = float(np.abs(best["Theta"] - rbest["Theta"]).max()) / max(1.0, float(np.abs(rbest["Theta"]).max()))
    if d_best > loose or set(best) != set(rbest):
        return tag, float("inf"), f"best_sol differs by {d_best:.2e} or in its keys {sorted(best)} vs {sorted(rbest)}"
    return tag, min(d_est, TOL), ""


def one_isolate(i, rng):
    """A grid in which some points' data are not finite (a NaN / Inf somewhere in their S, as a failed upstream estimate leaves
    it): those points end as 'solver error', and every OTHER point's result is what the grid without them returns (to rounding: the
    Omega-step's schedule is planned per batch, so the two runs differ in the last bits) -- the reference's sequential walk would
    lose the bad point only (helper/model_selection.py:619-633).  Latent on / off, compaction on / off."""
    from gglasso_amd import batch, synth
    p = int(rng.choice([q for q in P if 2 <= q <= 97]))
    single, K, reg = True, 1, "GGL"       # (the multiple-graph drivers share one S between the points: nothing to poison per point)
    G = int(rng.integers(2, 9))
    latent = bool(rng.random() < 0.5)
    lam1 = 10.0 ** rng.uniform(-2.0, 0.0, G)
    mu = 10.0 ** rng.uniform(-0.5, 0.7, G)
    tol = float(10.0 ** rng.uniform(-8, -6))
    compact = bool(rng.random() < 0.7)
    nbad = int(rng.integers(1, max(2, G // 2 + 1)))
    bad = np.sort(rng.choice(G, size=nbad, replace=False))
    good = np.array([g for g in range(G) if g not in set(bad.tolist())])
    tag = dict(i=i, kind="isolate", p=p, K=K, G=G, latent=latent, reg=reg if not single else "SGL", bad=bad.tolist(), compact=compact)
    if single:
        Sg = np.repeat(synth.make_problem("GGL", 1, p, seed=int(rng.integers(1 << 30)))[0], G, axis=0)        # (G,p,p), one S per point
        Sp = Sg.copy()
        for g in bad:
            a, b = int(rng.integers(p)), int(rng.integers(p))
            Sp[g, a, b] = Sp[g, b, a] = rng.choice([np.nan, np.inf, -np.inf])
        LAST.clear()
        LAST.update(S=Sp, lam1=lam1, mu=mu, latent=latent, tol=tol, compact=compact)
        kw = dict(Omega_0=np.eye(p), X_0=np.zeros((p, p)), tol=tol, rtol=tol, max_iter=300, latent=latent, compact=compact)
        res = quiet(batch.ADMM_SGL_batch, Sp, lam1, mu1=mu if latent else None, **kw)
        ref = quiet(batch.ADMM_SGL_batch, Sg[good], lam1[good], mu1=mu[good] if latent else None, **kw) if len(good) else []
    worst, note = 0.0, ""
    for g in bad:
        if res[g][1]["status"] != "solver error":
            return tag, float("inf"), f"poisoned point {g} ended as {res[g][1]['status']!r}"
    for n, g in enumerate(good):
        (sol, info), (rsol, rinfo) = res[g], ref[n]
        if info["iterations"] != rinfo["iterations"]:
            if abs(info["iterations"] - rinfo["iterations"]) > 1:
                return tag, float("inf"), f"point {g}: {info['iterations']} iterations vs {rinfo['iterations']}"
            note = f"iterations {info['iterations']} vs {rinfo['iterations']} at point {g}"
            continue
        if info["status"] != rinfo["status"]:
            return tag, float("inf"), f"point {g}: {info['status']!r} vs {rinfo['status']!r}"
        for nm in rsol:
            a, b = np.asarray(sol[nm]), np.asarray(rsol[nm])
            if not np.all(np.isfinite(a)):
                return tag, float("inf"), f"point {g}: {nm} not finite"
            worst = max(worst, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
    return tag, worst, note


def one_mgrid(i, rng):
    """grid_search (helper/model_selection.py:55-298) for the multiple-graph problems: the whole lambda1 x lambda2 grid as one
    batch with the device's statistics, against the same function walking the grid point by point with the criteria on the host
    and the ORACLE's ADMM_MGL as its solver -- started from the identity at every point like the batch (the reference's warm
    start, :224, is another path to the same optimum: it moves a point by up to 1e4 x the stopping tolerance where ADMM
    converges slowly, which would hide everything this comparison is for)."""
    from gglasso_amd import model_selection as ms, solver, synth
    from oracle import ggl_oracle as orc
    p = int(rng.choice([5, 8, 9, 15, 16, 17, 23, 31, 33, 47]))
    K = int(rng.integers(2, 5))
    reg = "GGL" if rng.random() < 0.5 else "FGL"
    n1, n2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    l1 = np.sort(10.0 ** rng.uniform(-1.5, -0.3, n1))[::-1].copy()
    use_w2 = bool(rng.random() < 0.5)
    l2 = None if use_w2 else np.sort(10.0 ** rng.uniform(-2.0, -0.5, n2))[::-1].copy()
    w2 = np.sort(rng.uniform(0.05, 0.6, n2)) if use_w2 else None
    latent = bool(rng.random() < 0.4)
    thr = bool(rng.random() < 0.3)
    method = "eBIC" if rng.random() < 0.6 else "AIC"
    gamma = float(rng.choice([0.1, 0.3, 0.5]))
    tol = float(10.0 ** rng.uniform(-9, -7))
    Nk = rng.integers(p, 4 * p + 5, K)
    S, _ = synth.make_problem(reg, K, p, N=int(Nk.max()), seed=int(rng.integers(1 << 30)))
    kw = dict(method=method, gamma=gamma, latent=latent, thresholding=thr, tol=tol, rtol=tol)
    if latent:
        nmu = int(rng.integers(1, 4))
        kw["mu_range"] = np.sort(10.0 ** rng.uniform(-0.3, 0.7, nmu))[::-1].copy()
        kw["ix_mu"] = rng.integers(0, nmu, (K, n1))          # one mu1 per instance and lambda1 column (model_selection.py:96-99)
    tag = dict(i=i, kind="mgrid", p=p, K=K, reg=reg, n1=n1, n2=n2, w2=use_w2, latent=latent, thresholding=thr, method=method,
               gamma=gamma, tol=tol)
    LAST.clear()
    LAST.update(S=S, N=Nk, l1=l1, l2=l2, w2=w2, reg=reg, **{k: v for k, v in kw.items()})
    st, ix, best = quiet(ms.grid_search, solver.ADMM_MGL, S, Nk, p, reg, l1, l2=l2, w2=w2, **kw)
    eyeK = np.repeat(np.eye(p)[None], K, axis=0)

    def cold(**a):
        a["Omega_0"] = eyeK.copy()
        return orc.ADMM_MGL(**a)

    rst, rix, rbest = quiet(ms.grid_search, cold, S, Nk, p, reg, l1, l2=l2, w2=w2, batched=False, **kw)
    loose = max(TOL, 100 * tol)            # (a point that stops one iteration apart is off by about its tolerance)
    if thr and not np.array_equal(st["TAU"], rst["TAU"]):
        return tag, 0.0, "iterations: another threshold chosen at a point"
    if not np.array_equal(st["SP"], rst["SP"]):
        return (tag,) + _pattern_note(st["AIC"], rst["AIC"])
    for nm_ in ["AIC"] + [("BIC", g) for g in sorted(rst["BIC"])]:
        a = st[nm_] if isinstance(nm_, str) else st[nm_[0]][nm_[1]]
        b = rst[nm_] if isinstance(nm_, str) else rst[nm_[0]][nm_[1]]
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            return tag, float("inf"), f"{nm_}: NaN pattern differs"
        ok = ~np.isnan(b)
        if ok.any() and float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(b[ok])))) > max(1e-6, 1e4 * tol):
            return tag, float("inf"), f"{nm_} differs by {float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(b[ok])))):.2e}"
    if latent and not np.array_equal(st["RANK"], rst["RANK"]):
        return tag, 0.0, "iterations: rank differs at a point (an eigenvalue of L at the cut to the solve's tolerance)"
    if tuple(np.atleast_1d(ix).tolist()) != tuple(np.atleast_1d(rix).tolist()):
        tab = rst["AIC"] if method == "AIC" else rst["BIC"][gamma]
        two = np.sort(tab[~np.isnan(tab)])[:2]
        if len(two) == 2 and abs(two[1] - two[0]) <= 1e-5 * max(1.0, abs(two[0])):
            return tag, 0.0, "iterations: best point differs between two scores that tie"
        return tag, float("inf"), f"best index {ix} vs {rix}"
    d = float(np.abs(best["Theta"] - rbest["Theta"]).max()) / max(1.0, float(np.abs(rbest["Theta"]).max()))
    if d > loose:
        return tag, float("inf"), f"best Theta differs by {d:.2e}"
    return tag, min(d, TOL), ""


def one_kgrid(i, rng):
    """K_single_grid (helper/model_selection.py:300-503): K independent single problems on one (lambda1, mu1) grid, all K x L x M
    points one batch -- against the host tables of the oracle's point-by-point solves and the selection rules (:443-466)
    restated here: best mu1 per instance and lambda1, then the best lambda1 uniformly and per instance."""
    from gglasso_amd import model_selection as ms, synth
    from oracle import ggl_oracle as orc
    p = int(rng.choice([q for q in P if 3 <= q <= 66]))
    K = int(rng.integers(2, 5))
    nl, latent = int(rng.integers(1, 5)), bool(rng.random() < 0.5)
    nm = int(rng.integers(1, 4)) if latent else 1
    lam = np.sort(10.0 ** rng.uniform(-1.8, -0.2, nl))[::-1].copy()
    mu = np.sort(10.0 ** rng.uniform(-0.3, 0.8, nm))[::-1].copy() if latent else None
    method = "eBIC" if rng.random() < 0.6 else "AIC"
    gamma = float(rng.choice([0.1, 0.3, 0.5]))
    thr = bool(rng.random() < 0.3)
    tol = float(10.0 ** rng.uniform(-9, -7))
    Nk = rng.integers(p, 4 * p + 5, K)
    S, _ = synth.make_problem("GGL", K, p, N=int(Nk.max()), seed=int(rng.integers(1 << 30)))
    tag = dict(i=i, kind="kgrid", p=p, K=K, nl=nl, nm=nm, latent=latent, method=method, gamma=gamma, thresholding=thr, tol=tol)
    LAST.clear()
    LAST.update(S=S, N=Nk, lam=lam, mu=mu, latent=latent, method=method, gamma=gamma, thr=thr, tol=tol)
    est_u, est_i, st = quiet(ms.K_single_grid, S, lam, Nk, method=method, gamma=gamma, latent=latent, mu_range=mu,
                             thresholding=thr, tol=tol, rtol=tol)
    mu_r = mu if latent else np.array([0])
    gammas = sorted(set(ms.DEFAULT_GAMMAS) | {gamma})
    eye = np.eye(p)
    tabs, ests, lows, bests = [], [], [], []
    for k in range(K):
        sols = []
        for j in range(nl):
            for m in range(nm):
                kw = dict(latent=True, mu1=float(mu[m])) if latent else {}
                sols.append(quiet(orc.ADMM_SGL, S[k], float(lam[j]), eye, X_0=eye, tol=tol, rtol=tol, max_iter=1000, **kw)[0])
        rb, re, rl, rs = ms._grid_tables(S[k], Nk[k], sols, None, lam, mu_r, latent, method, gamma, gammas, True, None, thr)
        tabs.append(rs); ests.append(re); lows.append(rl); bests.append(rb)
    if not np.array_equal(st["SP"], np.stack([t["SP"] for t in tabs])):
        if thr:                 # (another threshold chosen between two candidates that score alike moves the fit terms too)
            return tag, 0.0, "iterations: zero pattern or chosen threshold differs at a point"
        return (tag,) + _pattern_note(st["AIC"], np.stack([t["AIC"] for t in tabs]))
    for nm_, ref in (("AIC", np.stack([t["AIC"] for t in tabs])), ("BIC", np.stack([t["BIC"][gamma] for t in tabs]))):
        a = st[nm_]
        if not np.array_equal(np.isnan(a), np.isnan(ref)):
            return tag, float("inf"), f"{nm_}: NaN pattern differs"
        ok = ~np.isnan(ref)
        dev = float(np.max(np.abs(a[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok])))) if ok.any() else 0.0
        if dev > max(1e-7, 1e3 * tol):
            return tag, float("inf"), f"{nm_} differs by {dev:.2e}"
    if latent and not np.array_equal(st["RANK"], np.stack([t["RANK"] for t in tabs])):
        return tag, 0.0, "iterations: rank differs at a point (a singular value at numpy's cut)"
    # the selection (:443-466) from the REFERENCE tables; a tie (two scores within 1e-6) makes the index anybody's call
    table = np.stack([t["AIC"] if method == "AIC" else t["BIC"][gamma] for t in tabs])
    ix_mu = np.array([[np.nanargmin(table[k, j]) for j in range(nl)] for k in range(K)])
    score = np.take_along_axis(table, ix_mu[:, :, None], axis=2)[:, :, 0]

    def tied(v):
        w = np.sort(v[~np.isnan(v)])
        return len(w) > 1 and abs(w[1] - w[0]) <= 1e-6 * max(1.0, abs(w[0]))

    if any(tied(table[k, j]) for k in range(K) for j in range(nl)) or tied(score.sum(axis=0)) or any(tied(score[k]) for k in range(K)):
        return tag, 0.0, "iterations: a selection between two scores that tie"
    if not (np.array_equal(st["ix_mu"], ix_mu) and int(st["ix_uniform"]) == int(np.nanargmin(score.sum(axis=0)))
            and np.array_equal(st["ix_indv"], np.nanargmin(score, axis=1))):
        return tag, float("inf"), f"selection differs: ix_mu {st['ix_mu'].tolist()} ix_uniform {st['ix_uniform']} ix_indv {st['ix_indv'].tolist()}"
    loose = max(TOL, 100 * tol)
    ju = int(st["ix_uniform"])
    worst = 0.0
    for k in range(K):
        ru = ests[k][ju, ix_mu[k, ju]]
        worst = max(worst, float(np.abs(est_u["Theta"][k] - ru).max()) / max(1.0, float(np.abs(ru).max())),
                    float(np.abs(est_i["Theta"][k] - bests[k]["Theta"]).max()) / max(1.0, float(np.abs(bests[k]["Theta"]).max())))
        if latent:
            worst = max(worst, float(np.abs(est_u["L"][k] - lows[k][ju, ix_mu[k, ju]]).max()), float(np.abs(est_i["L"][k] - bests[k]["L"]).max()))
    if worst > loose:
        return tag, float("inf"), f"estimates differ by {worst:.2e}"
    return tag, min(worst, TOL), ""


def one_egrid(i, rng):
    """grid_search with ext_ADMM_MGL (instances of different dimension, helper/model_selection.py:55-298 with dict S and G): the
    grid as one padded batch (ext_ADMM_MGL_batch) against the point-by-point walk with the ORACLE's ext_ADMM_MGL from the identity
    start at every point."""
    from gglasso_amd import model_selection as ms
    from gglasso_amd.ext_solver import ext_ADMM_MGL
    from oracle import ggl_oracle as orc
    K = int(rng.integers(2, 5))
    universe = int(rng.choice([6, 9, 14, 20, 30]))
    sizes = [int(rng.integers(max(2, universe // 2), universe + 1)) for _ in range(K)]
    made = _nonconforming(rng, universe, sizes, prob=float(rng.uniform(0.05, 0.3)))
    tag = dict(i=i, kind="egrid", K=K, universe=universe, sizes=sizes)
    if made is None:
        return tag, 0.0, ""
    S, G, p = made
    n1, n2 = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    l1 = np.sort(10.0 ** rng.uniform(-1.5, -0.3, n1))[::-1].copy()
    l2 = np.sort(10.0 ** rng.uniform(-2.0, -0.5, n2))[::-1].copy()
    latent = bool(rng.random() < 0.4)
    method = "eBIC" if rng.random() < 0.6 else "AIC"
    gamma = float(rng.choice([0.1, 0.3, 0.5]))
    tol = float(10.0 ** rng.uniform(-8, -6))
    Nk = np.array([3 * n + 2 for n in p])
    kw = dict(method=method, gamma=gamma, latent=latent, tol=tol, rtol=tol, G=G)
    if latent:
        nmu = int(rng.integers(1, 3))
        kw["mu_range"] = np.sort(10.0 ** rng.uniform(-0.3, 0.7, nmu))[::-1].copy()
        kw["ix_mu"] = rng.integers(0, nmu, (K, n1))
    tag.update(n1=n1, n2=n2, latent=latent, method=method, gamma=gamma, tol=tol)
    LAST.clear()
    LAST.update(G=G, N=Nk, l1=l1, l2=l2, **{f"S_{k}": S[k] for k in range(K)}, **{k: v for k, v in kw.items() if k != "G"})

    def cold(**a):
        a["Omega_0"] = {k: np.eye(p[k]) for k in range(K)}
        return orc.ext_ADMM_MGL(**a)

    st, ix, best = quiet(ms.grid_search, ext_ADMM_MGL, S, Nk, p, "GGL", l1, l2=l2, **kw)
    rst, rix, rbest = quiet(ms.grid_search, cold, S, Nk, p, "GGL", l1, l2=l2, batched=False, **kw)
    if not np.array_equal(st["SP"], rst["SP"]):
        return (tag,) + _pattern_note(st["AIC"], rst["AIC"])
    for nm_ in ["AIC"] + [("BIC", g) for g in sorted(rst["BIC"])]:
        a = st[nm_] if isinstance(nm_, str) else st[nm_[0]][nm_[1]]
        b = rst[nm_] if isinstance(nm_, str) else rst[nm_[0]][nm_[1]]
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            return tag, float("inf"), f"{nm_}: NaN pattern differs"
        ok = ~np.isnan(b)
        dev = float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(b[ok])))) if ok.any() else 0.0
        if dev > max(1e-6, 1e4 * tol):
            return tag, float("inf"), f"{nm_} differs by {dev:.2e}"
    if latent and not np.array_equal(st["RANK"], rst["RANK"]):
        return tag, 0.0, "iterations: rank differs at a point"
    if tuple(np.atleast_1d(ix).tolist()) != tuple(np.atleast_1d(rix).tolist()):
        tab = rst["AIC"] if method == "AIC" else rst["BIC"][gamma]
        two = np.sort(tab[~np.isnan(tab)])[:2]
        if len(two) == 2 and abs(two[1] - two[0]) <= 1e-5 * max(1.0, abs(two[0])):
            return tag, 0.0, "iterations: best point differs between two scores that tie"
        return tag, float("inf"), f"best index {ix} vs {rix}"
    worst = max(float(np.abs(best["Theta"][k] - rbest["Theta"][k]).max()) / max(1.0, float(np.abs(rbest["Theta"][k]).max())) for k in range(K))
    if worst > max(TOL, 100 * tol):
        return tag, float("inf"), f"best Theta differs by {worst:.2e}"
    return tag, min(worst, TOL), ""


def one_bigbatch(i, rng):
    """A lambda (x mu) grid at the sizes where a batch runs as groups with their own Newton-Schulz schedules, concurrent parts,
    compaction and the C loop (p = 256 .. 500, 6 .. 20 points over two decades of lambda1: instances of very different
    conditioning) -- every point against the oracle's own solve; in half of the cases grouping is FORCED (GGL_OPT_GROUP_SCHED =
    13) through a subclass of the engine, so that the split is exercised whatever the size rule says."""
    from gglasso_amd import batch, solver, synth
    from oracle import ggl_oracle as orc
    p = int(rng.choice([256, 300, 333, 400, 500]))
    G = int(rng.integers(6, 21))
    latent = bool(rng.random() < 0.3)
    lam = 10.0 ** rng.uniform(-2.0, 0.0, G)
    mu = 10.0 ** rng.uniform(-0.3, 0.7, G)
    tol = float(10.0 ** rng.uniform(-8, -6))
    force = bool(rng.random() < 0.5)
    max_iter = int(rng.choice([25, 400]))
    S, _ = synth.make_problem("GGL", 1, p, N=int(rng.integers(p, 3 * p)), seed=int(rng.integers(1 << 30)))
    S = S[0]
    tag = dict(i=i, kind="bigbatch", p=p, G=G, latent=latent, tol=tol, forced_groups=force, max_iter=max_iter)
    LAST.clear()
    LAST.update(S=S, lam=lam, mu=mu, latent=latent, tol=tol, force=force, max_iter=max_iter)
    keep = solver.ENGINE
    seen = {}

    class Forced(keep):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.set_option("group_sched", 13 if force else 1)

        def close(self):
            try:
                gs = self.group_stats()
                seen["steps"] = seen.get("steps", 0) + gs["steps"]
            except Exception:
                pass
            return super().close()

    solver.ENGINE = Forced
    try:
        res = quiet(batch.ADMM_SGL_batch, S, lam, Omega_0=np.eye(p), X_0=np.zeros((p, p)), tol=tol, rtol=tol, max_iter=max_iter,
                    latent=latent, mu1=mu if latent else None)
    finally:
        solver.ENGINE = keep
    tag["grouped_steps"] = seen.get("steps")
    worst, note = 0.0, ""
    for g in range(G):
        kw = dict(latent=True, mu1=float(mu[g])) if latent else {}
        ref, rinfo = quiet(orc.ADMM_SGL, S, float(lam[g]), np.eye(p), X_0=np.zeros((p, p)), tol=tol, rtol=tol, max_iter=max_iter, **kw)
        sol, info = res[g]
        if info["iterations"] != rinfo["iterations"]:
            if abs(info["iterations"] - rinfo["iterations"]) > 1:
                return tag, float("inf"), f"point {g}: iterations {info['iterations']} vs {rinfo['iterations']}"
            note = f"point {g}: iterations {info['iterations']} vs {rinfo['iterations']}"
            continue
        if info["status"] != rinfo["status"]:
            return tag, float("inf"), f"point {g}: status {info['status']!r} vs {rinfo['status']!r}"
        for nm in ("Omega", "Theta", "X") + (("L",) if latent else ()):
            a, b = np.asarray(sol[nm]), np.asarray(ref[nm])
            if not np.all(np.isfinite(a)):
                return tag, float("inf"), f"point {g}: {nm} not finite"
            worst = max(worst, float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())))
    # (forced grouping still yields to the rules that keep a batch whole: one schedule for all when an instance needs the stable
    # iteration (kappa > 300), equal product counts, a K-sharded ctx -- so 'grouped_steps' is reported, not required)
    GROUPED.append((force, seen.get("steps", 0)))
    return tag, worst, note


def run_cases(cases, seed, out=print, dump_dir=None, big=True, kind="solver"):
    """Runs ``cases`` cases of the stream ``seed``; returns (cases off, last-bit stopping notes, largest deviation of the rest).
    ``big`` False keeps p <= 130 (the suite's quick pass); ``kind``: "solver" (one), "batch" (one_batch), "block" (one_block), "ext" (one_ext), "ops" (one_ops), "stats" (one_stats), "grid" (one_grid), "isolate" (one_isolate), "mgrid" (one_mgrid), "kgrid" (one_kgrid), "egrid" (one_egrid), "bigbatch" (one_bigbatch)."""
    global PBIG
    rng = np.random.default_rng(seed)
    keep, bad, notes, mx = PBIG, 0, 0, 0.0
    if not big:
        PBIG = P
    try:
        for i in range(cases):
            try:
                tag, worst, note = {"batch": one_batch, "block": one_block, "ext": one_ext, "ops": one_ops, "stats": one_stats, "grid": one_grid, "isolate": one_isolate, "mgrid": one_mgrid, "kgrid": one_kgrid, "egrid": one_egrid, "bigbatch": one_bigbatch}.get(kind, one)(i, rng)
            except Exception as e:                                      # a crash is a finding too
                out(f"case {i}: raised {type(e).__name__}: {e}")
                bad += 1
                continue
            if worst > TOL or (note and "exact-zero" not in note and "iterations" not in note):
                bad += 1
                out(f"OFF {worst:.3e} {note} {tag}")
                if dump_dir:
                    os.makedirs(dump_dir, exist_ok=True)
                    np.savez(os.path.join(dump_dir, f"case_seed{seed}_{i}.npz"), **{k: v for k, v in LAST.items() if v is not None})
            else:
                mx = max(mx, worst)
                if note:
                    notes += 1
                    out(f"note {worst:.3e} {note} {tag}")
    finally:
        PBIG = keep
    return bad, notes, mx
