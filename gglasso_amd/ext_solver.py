"""Drop-in ``ext_ADMM_MGL``: Group Graphical Lasso where the instances have DIFFERENT dimensions (some variables are
present in some instances only), with the reference's signature, asserts, status strings, prints and return dicts
(solver/ext_admm_solver.py:18-323 of fabian-sp/GGLasso; called by ``glasso_problem.solve()`` at problem.py:468 and as
the ``solver`` of ``grid_search``, helper/model_selection.py:628).

The reference walks Python dicts of (p_k,p_k) arrays instance by instance.  Here the K instances are ONE padded
(K,p,p) stack on the MI355X, p = max_k p_k: instance k is the leading (p_k,p_k) block of its slot, the rest of the
diagonal is an identity block (a decoupled fixed point of every step, include/ggl_hip.h), so the batched Omega-/L-step
kernels, the elementwise Theta-step and the gather/scatter group shrink over the bookkeeping array G all run once per
iteration for the whole problem.  Per iteration two scalars go down and five sums come back.
"""
import time
import warnings

import numpy as np

from . import solver as _solver
from .solver import as_c, residuals_from_norms


def check_G(G, p):
    """helper/ext_admm_helper.py:82-102 (same messages)."""
    K = G.shape[2]
    assert G.dtype == int, "G needs to be an integer array"
    assert np.all(G.sum(axis=2) >= -K), "G has rows with only -1 entries"
    assert np.all(((G == -1).sum(axis=0) == 2) | ((G == -1).sum(axis=0) == 0)), \
        "Only row or column index specified in some group"
    assert np.all((G[0, :, :] + G[1, :, :] == -2) | (G[0, :, :] != G[1, :, :])), "G has entries on the diagonal!"
    assert np.all(G >= -1), "No negative indices allowed (only -1 for indicating a missing feature)"
    assert np.all(G.max(axis=(0, 1)) < p), "indices larger as dimension were found"
    assert np.all(G[0, :, :] <= G[1, :, :]), "Only upper diagonal entries should be contained in G"


def _pad(blocks, K, p, P, identity):
    """dict k -> (p_k,p_k)  =>  (K,P,P) stack with an identity (or zero) block behind every instance."""
    out = np.zeros((K, P, P))
    for k in range(K):
        out[k, :p[k], :p[k]] = blocks[k]
        if identity and p[k] < P:
            d = np.arange(p[k], P)
            out[k, d, d] = 1.0
    return out


def _unpad(stack, K, p):
    return {k: stack[k, :p[k], :p[k]].copy() for k in range(K)}


def ext_ADMM_MGL(S, lambda1, lambda2, reg, Omega_0, G, X0=None, X1=None, tol=1e-5, rtol=1e-4,
                 stopping_criterion='boyd', rho=1., max_iter=1000, verbose=False, measure=False, latent=False, mu1=None):
    """Same arguments, defaults and return contract as the reference's ``ext_ADMM_MGL``: ``S``, ``Omega_0``, ``X0``,
    ``X1`` are dicts with keys 0..K-1 and (p_k,p_k) arrays, ``G`` the (2,L,K) integer bookkeeping array;
    ``sol = {'Omega','Theta','L','X0','X1'}`` (dicts again), ``info = {'status'}`` (+ ``runtime``, ``residual``).

    One restriction beyond ``check_G``: an entry (k,i,j) may belong to one group only (``create_group_array`` never
    produces anything else); the reference would shrink such an entry once per group, one group after the other."""
    K = len(S.keys())
    p = np.zeros(K, dtype=int)
    for k in np.arange(K):
        p[k] = S[k].shape[0]
    if isinstance(lambda1, float):
        lambda1 = lambda1 * np.ones(K)
    if latent:
        if isinstance(mu1, float):
            mu1 = mu1 * np.ones(K)
        assert mu1 is not None
        assert np.all(mu1 > 0)
        mu1 = as_c(mu1)
    else:
        mu1 = None
    lambda1 = as_c(lambda1)
    assert min(lambda1.min(), lambda2) > 0
    assert reg in ['GGL']
    check_G(G, p)
    assert rho > 0, "ADMM penalization parameter must be positive."
    assert stopping_criterion in ('boyd', 'kkt')

    P = int(p.max())
    Om0 = _pad(Omega_0, K, p, P, True)
    eng = _solver.ENGINE(_pad(S, K, p, P, True), Om0, Om0, _pad(X0, K, p, P, False) if X0 is not None
                         else np.zeros((K, P, P)))
    try:
        eng.ext_setup(p, G)
        eng.ext_set_state(Om0, None if X1 is None else _pad(X1, K, p, P, False))
        runtime = np.zeros(max_iter)
        residual = np.zeros(max_iter)
        status = ''
        dim = ((p ** 2 + p) / 2).sum()
        lambda2, rho = float(lambda2), float(rho)
        if verbose:
            print("------------ADMM Algorithm for Multiple Graphical Lasso----------------")
            if stopping_criterion == 'boyd':
                print("%4s\t%10s\t%10s\t%10s\t%10s" % ("iter", "r_t", "s_t", "eps_pri", "eps_dual"))
            else:
                print("%4s\t%10s" % ("iter", "kkt residual"))
        r_t = s_t = e_pri = e_dual = 0.0
        iter_t = -1
        for iter_t in range(max_iter):
            if measure:
                start = time.time()
            sq = eng.ext_step(rho, lambda1, lambda2, bool(latent), mu1)
            if measure:
                runtime[iter_t] = time.time() - start
            if stopping_criterion == 'boyd':
                r_t, s_t, e_pri, e_dual = residuals_from_norms(sq, rho, tol, rtol, dim)
                residual[iter_t] = max(r_t, s_t)
                if verbose:
                    print("%4d\t%10.4g\t%10.4g\t%10.4g\t%10.4g" % (iter_t, r_t, s_t, e_pri, e_dual))
                if (r_t <= e_pri) and (s_t <= e_dual):
                    status = 'optimal'
                    break
            else:
                eta_A = eng.ext_kkt(rho, lambda1, lambda2, bool(latent), mu1)
                residual[iter_t] = eta_A
                if verbose:
                    print("%4d\t%10.4g" % (iter_t, eta_A))
                if eta_A <= tol:
                    status = 'optimal'
                    break
        if status != 'optimal':
            if stopping_criterion == 'boyd':
                if r_t <= e_pri:
                    status = 'primal optimal'
                elif s_t <= e_dual:
                    status = 'dual optimal'
                else:
                    status = 'max iterations reached'
            else:
                status = 'max iterations reached'
        print(f"ADMM terminated after {iter_t+1} iterations with status: {status}.")

        if latent and hasattr(eng, "finalize_L"):
            eng.finalize_L()        # the returned L: one eigendecomposition where the L-steps were sign iterations
        # per-instance exit checks (ext_admm_solver.py:290-311): the decisions by batched Cholesky factorisations first, the
        # eigenvalues only if some instance fails one (solver._exit_report)
        rows = None
        if hasattr(eng, "exit_checks_fast"):
            f = eng.exit_checks_fast(bool(latent), 1e-5, 1e-5)
            if f[:, 3].min() > 0 and f[:, 4].min() > 0:
                rows = [(r[0], r[1], r[2], 1.0, 0.0) for r in f]
        for a_om, a_th, a_l, min_tl, min_l in (rows if rows is not None else eng.exit_checks_k(bool(latent))):
            for name, dev in (("Omega", a_om), ("Theta", a_th), ("L", a_l)):
                if dev > 1e-5:
                    warnings.warn(f"{name} variable is not symmetric, largest deviation is {dev}.")
            if min_tl <= 1e-5:
                print("WARNING: Theta (Theta-L resp.) may be not positive definite -- increase accuracy!")
            if latent and min_l <= -1e-5:
                print("WARNING: L may be not positive semidefinite -- increase accuracy!")
        st, xs = eng.state(), eng.ext_state()
    finally:
        eng.close()
    sol = {'Omega': _unpad(st['Omega'], K, p), 'Theta': _unpad(st['Theta'], K, p), 'L': _unpad(st['L'], K, p),
           'X0': _unpad(st['X'], K, p), 'X1': _unpad(xs['X1'], K, p)}
    if measure:
        info = {'status': status, 'runtime': runtime[:iter_t + 1], 'residual': residual[:iter_t + 1]}
    else:
        info = {'status': status}
    return sol, info


def ext_ADMM_MGL_batch(S, lambda1, lambda2, reg, G, tol=1e-5, rtol=1e-4, rho=1., max_iter=1000, latent=False, mu1=None,
                       verbose=False):
    """``ext_ADMM_MGL(S, lambda1[g], lambda2[g], reg, Omega_0 = identity, G, ...)`` for every g of the 1-D arrays
    ``lambda1`` / ``lambda2`` at once -- the grid points the MAIN LOOP of the reference's ``grid_search``
    (helper/model_selection.py:208-224) hands to ``ext_ADMM_MGL`` one after the other.  The points are independent
    problems on the same data and the same bookkeeping array G: they live as the slabs of ONE padded (ngrid*K, P, P)
    stack, one batched Omega-step (and L-step) over all ngrid*K matrices, one Theta-step, one group shrink (with a
    grid-point dimension) and one dual update per iteration.  The solver has no rho update (ext_admm_solver.py), so all
    points share rho; every point keeps its own residuals and stopping iteration and is snapshotted when it converges
    (it keeps iterating harmlessly until the batch is done); a point whose data turn non-finite ends with status
    'solver error' and is parked, the others go on.  lambda1[g]: scalar or (K,); mu1: (K,) shared or (ngrid,K).
    Returns a list of ``(sol, info)`` as ``ext_ADMM_MGL`` (Boyd criterion), ``info`` with 'iterations' added."""
    K = len(S.keys())
    p = np.array([S[k].shape[0] for k in range(K)], dtype=int)
    lam2 = as_c(np.atleast_1d(lambda2)).reshape(-1)
    ng = len(lam2)
    lam1 = np.stack([np.broadcast_to(np.asarray(l, dtype=np.float64), (K,)) for l in lambda1]) if not np.isscalar(lambda1) \
        else np.full((ng, K), float(lambda1))
    assert lam1.shape == (ng, K)
    assert min(lam1.min(), lam2.min()) > 0
    assert reg in ['GGL']
    check_G(G, p)
    assert rho > 0, "ADMM penalization parameter must be positive."
    if latent:
        assert mu1 is not None
        mu = np.asarray(mu1, dtype=np.float64)
        if mu.ndim == 0:
            mu = mu * np.ones(K)
        mu = as_c(np.broadcast_to(mu, (ng, K))).reshape(-1)
        assert np.all(mu > 0)
    else:
        mu = None
    P = int(p.max())
    Sp = _pad(S, K, p, P, True)
    Om0 = _pad({k: np.eye(p[k]) for k in range(K)}, K, p, P, True)
    rep = lambda A: as_c(np.broadcast_to(A, (ng,) + A.shape)).reshape(ng * K, P, P)
    eng = _solver.ENGINE(rep(Sp), rep(Om0), rep(Om0), np.zeros((ng * K, P, P)))
    try:
        eng.ext_setup_batch(ng, p, G)
        eng.ext_set_state(rep(Om0), None)
        if hasattr(eng, "set_option"):
            eng.set_option("isolate", 1)        # a point with non-finite data costs that point only (batch.ADMM_MGL_batch)
        dim = ((p ** 2 + p) / 2).sum()
        rho = float(rho)
        lam1f = as_c(lam1.reshape(-1))
        done = np.zeros(ng, dtype=bool)
        results = [None] * ng
        last = [None] * ng

        def collect(g, status, iters):
            parts = [eng.state_k(g * K + k, True) for k in range(K)]
            xs = eng.ext_state()
            cut = lambda A, k: np.ascontiguousarray(A[:p[k], :p[k]])
            sol = {'Omega': {k: cut(parts[k]['Omega'], k) for k in range(K)},
                   'Theta': {k: cut(parts[k]['Theta'], k) for k in range(K)},
                   'L': {k: cut(parts[k]['L'], k) for k in range(K)},
                   'X0': {k: cut(parts[k]['X'], k) for k in range(K)},
                   'X1': {k: cut(xs['X1'][g * K + k], k) for k in range(K)}}
            results[g] = (sol, {'status': status, 'iterations': iters})
            if latent:
                for k in range(K):
                    eng.snapshot_k(g * K + k)

        for it in range(max_iter):
            sq = eng.ext_batch_step(ng, rho, lam1f, lam2, bool(latent), mu)
            for g in range(ng):
                if done[g]:
                    continue
                if not np.all(np.isfinite(sq[g])):
                    collect(g, 'solver error', it + 1)
                    done[g] = True
                    if hasattr(eng, "reset_instance"):
                        for k in range(K):
                            eng.reset_instance(g * K + k)
                    continue
                r_t, s_t, e_pri, e_dual = residuals_from_norms(sq[g], rho, tol, rtol, dim)
                last[g] = (r_t, s_t, e_pri, e_dual)
                if verbose:
                    print("%4d\t%3d\t%10.4g\t%10.4g\t%10.4g\t%10.4g" % (it, g, r_t, s_t, e_pri, e_dual))
                if (r_t <= e_pri) and (s_t <= e_dual):
                    done[g] = True
                    collect(g, 'optimal', it + 1)
            if done.all():
                break
        for g in range(ng):
            if results[g] is None:
                r_t, s_t, e_pri, e_dual = last[g]
                status = 'primal optimal' if r_t <= e_pri else ('dual optimal' if s_t <= e_dual
                                                                else 'max iterations reached')
                collect(g, status, max_iter)
        if latent:
            # every problem's L_k as one eigendecomposition of its last L-step's input (HipEngine.finalize_L)
            _, rk = eng.finalize_L(1)
            for g in range(ng):
                for k in range(K):
                    if rk[g * K + k] >= 0:
                        results[g][0]['L'][k] = np.ascontiguousarray(eng.snapshot_L_k(g * K + k)[:p[k], :p[k]])
    finally:
        eng.close()
    return results
