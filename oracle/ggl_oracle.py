"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A NumPy restatement of the ADMM hot path of fabian-sp/GGLasso (reference @ /root/reference,
v0.2.1).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module; ``gglasso_amd`` never does (the product path fails loudly without the HIP
library instead of falling back to anything in here).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference in the build
container and stores inputs + reference outputs in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors (operators <= 1e-12, 10-iteration trajectories
<= 1e-10, converged Theta <= 1e-8) plus the reference's own known-answer test (lambda1_mask = 0
=> Theta = inv(S), tests/test_solvers.py:191-216).  Round 6: also on DRAWN inputs -- ``tests/golden/fuzz_oracle_vs_reference.py``
runs the real reference against this module with the case generators of ``tests/fuzz_checks.py`` (ADMM_MGL / ADMM_SGL, block_SGL,
ext_ADMM_MGL, the operators on engineered spectra; 9 040 cases, <= 3e-12; profiles/r6_oracle_pinned_on_random_inputs.txt).

Third-party arithmetic: the reference calls ``numpy.linalg.eigh`` (LAPACK dsyevd via NumPy; pin
``numpy>=1.17.3,<2.0.0`` in pyproject.toml:37, NumPy 2.2.6/OpenBLAS 0.3.29 in this image) at
admm_solver.py:181,199 and single_admm_solver.py:164,174.  The oracle calls the same routine, so the
only thing restated is what the reference does around it.  Eigenvectors are never compared (sign /
order / degenerate-subspace ambiguity); only Q f(D) Q^T is.

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/gglasso/).
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def _clib():
    """C restatement of prox_p (oracle/ggl_oracle.c), used for sizes where Python loops are too slow."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "libggl_oracle.so")
        if not os.path.exists(path):
            return None
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.oracle_prox_p.argtypes = [dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int]
        lib.oracle_prox_p.restype = ctypes.c_int
        lib.oracle_condat.argtypes = [dp, dp, ctypes.c_int, ctypes.c_double]
        lib.oracle_condat.restype = None
        _CLIB = lib
    return _CLIB


# ---------------------------------------------------------------------------------------------
# elementwise / group / fused prox operators
# ---------------------------------------------------------------------------------------------

def prox_1norm(v, l):
    """solver/ggl_helper.py:12-14 -- soft threshold."""
    return np.sign(v) * np.maximum(np.abs(v) - l, 0.)


def prox_od_1norm(A, l):
    """solver/ggl_helper.py:16-27 -- soft threshold everywhere, diagonal restored from the input.
    ``l`` is a scalar or a (p,p) array (lambda1 * lambda1_mask, single_admm_solver.py:114)."""
    res = np.sign(A) * np.maximum(np.abs(A) - l, 0.)
    d = np.arange(min(A.shape))
    res[d, d] = A[d, d]
    return res


def prox_2norm(v, l):
    """solver/ggl_helper.py:38-43 -- prox of the Euclidean (Frobenius) norm."""
    a = np.maximum(np.linalg.norm(v), l)
    return v * (a - l) / a


def prox_phi_ggl(v, l1, l2):
    """solver/ggl_helper.py:68-71."""
    return prox_2norm(prox_1norm(v, l1), l2)


def condat_method(y, lam):
    """solver/fgl_helper.py:11-68 -- Condat's direct 1-D total-variation prox (pure-Python loop;
    small cases only, the C twin in ggl_oracle.c handles the rest)."""
    y = np.asarray(y, dtype=np.float64)
    N = len(y)
    x = np.zeros(N)
    k = k0 = kplus = kminus = 0
    vmin = y[0] - lam
    vmax = y[0] + lam
    umin = lam
    umax = -lam
    while True:
        while k == N - 1:
            if umin < 0:
                x[k0:kminus + 1] = vmin
                kminus += 1
                k = k0 = kminus
                umin = lam
                vmin = y[k]
                umax = y[k] + lam - vmax
            elif umax > 0:
                x[k0:kplus + 1] = vmax
                kplus += 1
                k = k0 = kplus
                umax = -lam
                vmax = y[k]
                umin = y[k] - lam - vmin
            else:
                x[k0:] = vmin + umin / (k - k0 + 1)
                return x
            if k == N - 1:
                x[k] = vmin + umin
                return x
        if y[k + 1] + umin - vmin < -lam:
            x[k0:kminus + 1] = vmin
            kminus += 1
            k = kplus = k0 = kminus
            vmin = y[k]
            vmax = y[k] + 2 * lam
            umin = lam
            umax = -lam
        elif y[k + 1] + umax - vmax > lam:
            x[k0:kplus + 1] = vmax
            kplus += 1
            k = kminus = k0 = kplus
            vmin = y[k] - 2 * lam
            vmax = y[k]
            umin = lam
            umax = -lam
        else:
            k += 1
            umin = umin + y[k] - vmin
            umax = umax + y[k] - vmax
            if umin >= lam:
                vmin += (umin - lam) / (k - k0 + 1)
                umin = lam
                kminus = k
            if umax <= -lam:
                vmax += (umax + lam) / (k - k0 + 1)
                umax = -lam
                kplus = k


def prox_tv(v, l):
    """solver/ggl_helper.py:126-129."""
    return condat_method(v, l)


def prox_phi_fgl(v, l1, l2):
    """solver/ggl_helper.py:131-134 -- TV prox first, then soft threshold."""
    return prox_1norm(prox_tv(v, l2), l1)


def prox_p_loops(X, l1, l2, reg):
    """solver/ggl_helper.py:190-207 restated pair by pair (pure Python; small cases)."""
    assert np.abs(X - X.transpose(0, 2, 1)).max() <= 1e-5, "input X is not symmetric"
    assert min(l1, l2) > 0
    assert reg in ('GGL', 'FGL')
    K, p, _ = X.shape
    M = np.zeros((K, p, p))
    for i in range(p):
        for j in range(i, p):
            if i == j:
                M[:, i, j] = 0.5 * X[:, i, j]
            elif reg == 'GGL':
                M[:, i, j] = prox_phi_ggl(X[:, i, j], l1, l2)
            else:
                M[:, i, j] = prox_phi_fgl(X[:, i, j], l1, l2)
    return M + M.transpose(0, 2, 1)


def prox_p(X, l1, l2, reg):
    """solver/ggl_helper.py:190-207.  Upper triangle (i<j) decides, output is exactly symmetric,
    the diagonal passes through.  GGL is vectorised over all pairs; FGL uses the C restatement of
    Condat's scan when it is built, the Python loop otherwise."""
    assert np.abs(X - X.transpose(0, 2, 1)).max() <= 1e-5, "input X is not symmetric"
    assert min(l1, l2) > 0, "lambda 1 and lambda2 have to be positive"
    assert reg in ('GGL', 'FGL')
    K, p, _ = X.shape
    if reg == 'GGL':
        iu = np.triu_indices(p, 1)
        U = prox_1norm(X[:, iu[0], iu[1]], l1)                  # (K, npairs)
        a = np.maximum(np.sqrt(np.sum(U * U, axis=0)), l2)
        U = U * ((a - l2) / a)
        M = np.zeros((K, p, p))
        M[:, iu[0], iu[1]] = U
        M = M + M.transpose(0, 2, 1)
        d = np.arange(p)
        M[:, d, d] = X[:, d, d]
        return M
    lib = _clib()
    if lib is None:
        return prox_p_loops(X, l1, l2, reg)
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    out = np.empty_like(Xc)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.oracle_prox_p(Xc.ctypes.data_as(dp), out.ctypes.data_as(dp), K, p, l1, l2, 2)
    assert rc == 0
    return out


def prox_p_c(X, l1, l2, reg):
    """prox_p through the C restatement for both penalties (cross-check of the two oracles)."""
    lib = _clib()
    assert lib is not None, "oracle/libggl_oracle.so not built (run oracle/build.sh)"
    K, p, _ = X.shape
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    out = np.empty_like(Xc)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.oracle_prox_p(Xc.ctypes.data_as(dp), out.ctypes.data_as(dp), K, p, l1, l2,
                           1 if reg == 'GGL' else 2)
    assert rc == 0
    return out


def P_val(X, l1, l2, reg):
    """solver/ggl_helper.py:162-176 -- value of the regulariser (off-diagonal only)."""
    K, p, _ = X.shape
    iu = np.triu_indices(p, 1)
    V = X[:, iu[0], iu[1]]
    res = l1 * np.abs(V).sum()
    if reg == 'GGL':
        res += l2 * np.sqrt((V * V).sum(axis=0)).sum()
    else:
        res += l2 * np.abs(V[1:] - V[:-1]).sum()
    return 2 * res


# ---------------------------------------------------------------------------------------------
# log-det prox and low-rank prox (eigenvalue maps)
# ---------------------------------------------------------------------------------------------

def phip(d, beta):
    """solver/ggl_helper.py:272-274."""
    return 0.5 * (np.sqrt(d ** 2 + 4 * beta) + d)


def phiplus(beta, D, Q):
    """solver/ggl_helper.py:280-303 -- B = Q diag(phip(D)) Q^T."""
    return (Q * phip(D, beta)) @ Q.T


def prox_rank_norm(A, beta, D=np.array([]), Q=np.array([])):
    """solver/ggl_helper.py:29-36 -- B = Q diag(max(D-beta,0)) Q^T (eigh recomputed if D is absent)."""
    if len(D) != A.shape[0]:
        D, Q = np.linalg.eigh(A)
    return (Q * np.maximum(D - beta, 0.)) @ Q.T


def phiplus_stack(W, beta):
    """Omega-step of admm_solver.py:180-187 for a whole (K,p,p) stack; beta scalar or (K,)."""
    D, Q = np.linalg.eigh(W)
    beta = np.broadcast_to(np.asarray(beta, dtype=np.float64), (W.shape[0],))
    return (Q * phip(D, beta[:, None])[:, None, :]) @ Q.transpose(0, 2, 1), D


def rank_stack(C, beta):
    """L-step of admm_solver.py:197-205 for a whole stack."""
    D, Q = np.linalg.eigh(C)
    beta = np.broadcast_to(np.asarray(beta, dtype=np.float64), (C.shape[0],))
    return (Q * np.maximum(D - beta[:, None], 0.)[:, None, :]) @ Q.transpose(0, 2, 1)


def f_obj(Omega, S):
    """solver/ggl_helper.py:266-270 -- sum_k -log det Omega_k + <Omega, S>."""
    return (-np.log(np.linalg.det(Omega))).sum() + np.sum(Omega * S)


# ---------------------------------------------------------------------------------------------
# stopping criteria
# ---------------------------------------------------------------------------------------------

def ADMM_stopping_criterion(Omega, Omega_t_1, Theta, L, X, S, rho, eps_abs, eps_rel, latent=False):
    """solver/admm_solver.py:316-331 (K,p,p) and solver/single_admm_solver.py:277-291 (p,p).
    Note dim*eps_abs with dim = K(p^2+p)/2, not sqrt(dim)."""
    if S.ndim == 3:
        K, p, _ = S.shape
    else:
        K, p = 1, S.shape[0]
    dim = K * ((p ** 2 + p) / 2)
    e_pri = dim * eps_abs + eps_rel * np.maximum(np.linalg.norm(Omega), np.linalg.norm(Theta - L))
    e_dual = dim * eps_abs + eps_rel * rho * np.linalg.norm(X)
    r = np.linalg.norm(Omega - Theta + L)
    s = rho * np.linalg.norm(Omega - Omega_t_1)
    return r, s, e_pri, e_dual


def kkt_stopping_criterion_mgl(Omega, Theta, L, X, S, lambda1, lambda2, nk, reg, latent=False, mu1=None):
    """solver/admm_solver.py:333-371 (X is the UNscaled dual here: the caller passes rho*X)."""
    term1 = np.linalg.norm(Theta - prox_p(Theta + X, lambda1, lambda2, reg)) / (1 + np.linalg.norm(Theta))
    term2 = np.linalg.norm(Theta - Omega - L) / (1 + np.linalg.norm(Theta))
    proxK, _ = phiplus_stack(Omega - nk * S - X, nk[:, 0, 0])
    term3 = np.linalg.norm(Omega - proxK) / (1 + np.linalg.norm(Omega))
    term4 = 0
    if latent:
        proxL = rank_stack(L - X, mu1)
        term4 = np.linalg.norm(L - proxL) / (1 + np.linalg.norm(L))
    return max(term1, term2, term3, term4)


def kkt_stopping_criterion_sgl(Omega, Theta, L, X, S, lambda1, latent=False, mu1=None):
    """solver/single_admm_solver.py:293-320."""
    term1 = np.linalg.norm(Theta - prox_od_1norm(Theta + X, l=lambda1)) / (1 + np.linalg.norm(Theta))
    term2 = np.linalg.norm(Omega - Theta + L) / (1 + np.linalg.norm(Theta))
    D, Q = np.linalg.eigh(Omega - S - X)
    term3 = np.linalg.norm(Omega - phiplus(1, D, Q)) / (1 + np.linalg.norm(Omega))
    term4 = 0
    if latent:
        D, Q = np.linalg.eigh(L - X)
        term4 = np.linalg.norm(L - prox_rank_norm(L - X, mu1, D, Q)) / (1 + np.linalg.norm(L))
    return max(term1, term2, term3, term4)


# ---------------------------------------------------------------------------------------------
# full host loops (control flow of the reference, operators from above)
# ---------------------------------------------------------------------------------------------

def ADMM_MGL(S, lambda1, lambda2, reg, Omega_0, Theta_0=np.array([]), X_0=np.array([]),
             n_samples=None, tol=1e-5, rtol=1e-4, stopping_criterion='boyd', update_rho=True,
             rho=1., max_iter=1000, verbose=False, measure=False, latent=False, mu1=None,
             history=None):
    """solver/admm_solver.py:13-313.  ``history`` (a list) receives (r,s,e_pri,e_dual,rho) per
    iteration; it is an oracle-only extra for trajectory tests."""
    assert Omega_0.shape == S.shape
    assert S.shape[1] == S.shape[2]
    assert reg in ['GGL', 'FGL']
    assert min(lambda1, lambda2) > 0
    K, p, _ = S.shape
    assert rho > 0
    if latent:
        if isinstance(mu1, float):
            mu1 = mu1 * np.ones(K)
        assert mu1 is not None
        assert np.all(mu1 > 0)
    if n_samples is None:
        nk = np.ones((K, 1, 1))
    elif isinstance(n_samples, int):
        nk = n_samples * np.ones((K, 1, 1))
    else:
        nk = np.asarray(n_samples, dtype=np.float64).reshape(K, 1, 1)

    Omega_t = Omega_0.copy()
    if len(Theta_0) == 0:
        Theta_0 = Omega_0.copy()
    if len(X_0) == 0:
        X_0 = np.zeros((K, p, p))
    Theta_t = Theta_0.copy()
    L_t = np.zeros((K, p, p))
    X_t = X_0.copy()
    residual = np.zeros(max_iter)
    objective = np.zeros(max_iter)
    status = ''

    for iter_t in range(max_iter):
        Omega_t_1 = Omega_t
        W_t = Theta_t - L_t - X_t - (nk / rho) * S
        Omega_t, _ = phiplus_stack(W_t, nk[:, 0, 0] / rho)
        Theta_t = prox_p(Omega_t + L_t + X_t, (1 / rho) * lambda1, (1 / rho) * lambda2, reg)
        if latent:
            L_t = rank_stack(Theta_t - X_t - Omega_t, mu1 / rho)
        X_t = X_t + Omega_t - Theta_t + L_t
        if measure:
            objective[iter_t] = f_obj(Omega_t, S) + P_val(Theta_t, lambda1, lambda2, reg)

        if stopping_criterion == 'boyd':
            r_t, s_t, e_pri, e_dual = ADMM_stopping_criterion(Omega_t, Omega_t_1, Theta_t, L_t, X_t,
                                                              S, rho, tol, rtol, latent)
            if history is not None:
                history.append((r_t, s_t, e_pri, e_dual, rho))
            if update_rho:
                if r_t >= 10 * s_t:
                    rho_new = 2 * rho
                elif s_t >= 10 * r_t:
                    rho_new = 0.5 * rho
                else:
                    rho_new = 1. * rho
                X_t = (rho / rho_new) * X_t
                rho = rho_new
            residual[iter_t] = max(r_t, s_t)
            if (r_t <= e_pri) and (s_t <= e_dual):
                status = 'optimal'
                break
        else:
            eta_A = kkt_stopping_criterion_mgl(Omega_t, Theta_t, L_t, rho * X_t, S, lambda1, lambda2,
                                               nk, reg, latent, mu1)
            residual[iter_t] = eta_A
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

    sol = {'Omega': Omega_t, 'Theta': Theta_t, 'L': L_t, 'X': X_t}
    info = {'status': status, 'iterations': iter_t + 1, 'rho': rho}
    if measure:
        info['residual'] = residual[:iter_t + 1]
        info['objective'] = objective[:iter_t + 1]
    return sol, info


def ADMM_SGL(S, lambda1, Omega_0, Theta_0=np.array([]), X_0=np.array([]), rho=1., max_iter=1000,
             tol=1e-7, rtol=1e-4, stopping_criterion='boyd', update_rho=True, verbose=False,
             measure=False, latent=False, mu1=None, lambda1_mask=None, history=None):
    """solver/single_admm_solver.py:15-275."""
    assert Omega_0.shape == S.shape
    assert S.shape[0] == S.shape[1]
    p = S.shape[0]
    assert lambda1 > 0
    if lambda1_mask is not None:
        assert lambda1_mask.shape == (p, p)
        assert np.all(lambda1_mask >= 0)
        assert np.all(np.abs(lambda1_mask.T - lambda1_mask) <= 1e-5)
        lambda1 = lambda1 * lambda1_mask
    assert np.all(lambda1 >= 0)
    assert stopping_criterion in ["boyd", "kkt"]
    if latent:
        assert mu1 is not None
        assert mu1 > 0
    assert rho > 0

    Omega_t = Omega_0.copy()
    if len(Theta_0) == 0:
        Theta_0 = Omega_0.copy()
    if len(X_0) == 0:
        X_0 = np.zeros((p, p))
    Theta_t = Theta_0.copy()
    L_t = np.zeros((p, p))
    X_t = X_0.copy()
    residual = np.zeros(max_iter)
    status = ''

    for iter_t in range(max_iter):
        W_t = Theta_t - L_t - X_t - (1 / rho) * S
        eigD, eigQ = np.linalg.eigh(W_t)
        Omega_t_1 = Omega_t
        Omega_t = phiplus(beta=1 / rho, D=eigD, Q=eigQ)
        Theta_t = prox_od_1norm(Omega_t + L_t + X_t, (1 / rho) * lambda1)
        if latent:
            C_t = Theta_t - X_t - Omega_t
            eigD1, eigQ1 = np.linalg.eigh(C_t)
            L_t = prox_rank_norm(C_t, mu1 / rho, D=eigD1, Q=eigQ1)
        X_t = X_t + Omega_t - Theta_t + L_t

        if stopping_criterion == 'boyd':
            r_t, s_t, e_pri, e_dual = ADMM_stopping_criterion(Omega_t, Omega_t_1, Theta_t, L_t, X_t,
                                                              S, rho, tol, rtol, latent)
            if history is not None:
                history.append((r_t, s_t, e_pri, e_dual, rho))
            if update_rho:
                if r_t >= 10 * s_t:
                    rho_new = 2 * rho
                elif s_t >= 10 * r_t:
                    rho_new = 0.5 * rho
                else:
                    rho_new = 1. * rho
                X_t = (rho / rho_new) * X_t
                rho = rho_new
            residual[iter_t] = max(r_t, s_t)
            if (r_t <= e_pri) and (s_t <= e_dual):
                status = 'optimal'
                break
        else:
            eta_A = kkt_stopping_criterion_sgl(Omega_t, Theta_t, L_t, rho * X_t, S, lambda1, latent, mu1)
            residual[iter_t] = eta_A
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

    if latent:
        sol = {'Omega': Omega_t, 'Theta': Theta_t, 'L': L_t, 'X': X_t}
    else:
        sol = {'Omega': Omega_t, 'Theta': Theta_t, 'X': X_t}
    info = {'status': status, 'iterations': iter_t + 1, 'rho': rho}
    if measure:
        info['residual'] = residual[:iter_t + 1]
    return sol, info


def get_connected_components(S, lambda1):
    """solver/single_admm_solver.py:478-490 -- components of the graph |S_ij| > lambda1_ij (diagonal kept)."""
    from scipy.sparse.csgraph import connected_components
    A = (np.abs(S) > lambda1).astype(int)
    np.fill_diagonal(A, 1)
    numC, labels = connected_components(A, directed=False, return_labels=True)
    return numC, [np.flatnonzero(labels == i) for i in range(numC)]


def block_SGL(S, lambda1, Omega_0, Theta_0=None, X_0=None, rho=1., max_iter=1000, tol=1e-7, rtol=1e-3,
              stopping_criterion="boyd", update_rho=True, lambda1_mask=None):
    """solver/single_admm_solver.py:326-475 -- Witten/Friedman/Simon block splitting: singletons in closed
    form 1/S_ii (off-diagonal penalty, :432-438), ADMM_SGL per block (:445-459), reassembly through the
    inverse permutation (:466-473).  Returns only ``sol`` like the reference."""
    from scipy.linalg import block_diag
    p = S.shape[0]
    if lambda1_mask is None:
        lambda1_mask = np.ones((p, p))
    if Theta_0 is None:
        Theta_0 = Omega_0.copy()
    if X_0 is None:
        X_0 = np.zeros((p, p))
    numC, allC = get_connected_components(S, lambda1 * lambda1_mask)
    allOmega, allTheta, allX = [], [], []
    for C in allC:
        if len(C) == 1:
            v = 1 / S[C, C]
            allOmega.append(v); allTheta.append(v); allX.append(np.array([0]))
        else:
            ix = np.ix_(C, C)
            bs, _ = ADMM_SGL(S[ix], lambda1, Omega_0[ix], Theta_0=Theta_0[ix], X_0=X_0[ix], tol=tol, rtol=rtol,
                             stopping_criterion=stopping_criterion, update_rho=update_rho, rho=rho,
                             max_iter=max_iter, lambda1_mask=lambda1_mask[ix])
            allOmega.append(bs['Omega']); allTheta.append(bs['Theta']); allX.append(bs['X'])
    per = np.hstack(allC)
    inv = np.empty_like(per)
    inv[per] = np.arange(per.size)
    ixp = np.ix_(inv, inv)
    return {'Omega': block_diag(*allOmega)[ixp], 'Theta': block_diag(*allTheta)[ixp], 'X': block_diag(*allX)[ixp]}


# ---------------------------------------------------------------------------------------------
# ext_ADMM_MGL: Group Graphical Lasso over instances of DIFFERENT dimension (solver/ext_admm_solver.py)
# ---------------------------------------------------------------------------------------------

def check_G(G, p):
    """helper/ext_admm_helper.py:82-102."""
    K = G.shape[2]
    assert G.dtype == int, "G needs to be an integer array"
    assert np.all(G.sum(axis=2) >= -K), "G has rows with only -1 entries"
    assert np.all(((G == -1).sum(axis=0) == 2) | ((G == -1).sum(axis=0) == 0)), \
        "Only row or column index specified in some group"
    assert np.all((G[0] + G[1] == -2) | (G[0] != G[1])), "G has entries on the diagonal!"
    assert np.all(G >= -1), "No negative indices allowed (only -1 for indicating a missing feature)"
    assert np.all(G.max(axis=(0, 1)) < p), "indices larger as dimension were found"
    assert np.all(G[0] <= G[1]), "Only upper diagonal entries should be contained in G"


def construct_trivial_G(p, K):
    """helper/ext_admm_helper.py:31-44: every pair i<j present in every instance, row-major pair order."""
    iu = np.triu_indices(p, 1)
    G = np.zeros((2, len(iu[0]), K), dtype=int)
    G[0] = iu[0][:, None]
    G[1] = iu[1][:, None]
    return G


def prox_2norm_G(X, G, l2):
    """solver/ext_admm_solver.py:394-453 (prox_2norm_G + prox_G_inner): for every group l the vector of the entries
    X[k][G[0,l,k], G[1,l,k]] of the instances that hold the pair (G != -1) is shrunk with threshold
    l2*sqrt(group size) and written back to (i,j) and (j,i); entries outside every group pass through.
    X: dict k -> (p_k,p_k) symmetric array.  The groups are processed in order on the same arrays, exactly as the
    reference's loop does (an entry listed in two groups is shrunk twice)."""
    assert l2 > 0
    K = len(X)
    for k in range(K):
        assert np.abs(X[k] - X[k].T).max() <= 1e-5, "X[k] has to be symmetric"
    assert G.shape[0] == 2 and G.shape[2] == K
    group_size = (G[0] != -1).sum(axis=1)
    out = {k: X[k].copy() for k in range(K)}
    for l in range(G.shape[1]):
        ks = np.flatnonzero(G[0, l] != -1)
        v = np.array([out[k][G[0, l, k], G[1, l, k]] for k in ks])
        lam = l2 * np.sqrt(group_size[l])
        a = max(np.sqrt((v ** 2).sum()), lam)
        z = v * (a - lam) / a
        for n, k in enumerate(ks):
            out[k][G[0, l, k], G[1, l, k]] = z[n]
            out[k][G[1, l, k], G[0, l, k]] = z[n]
    return out


def _G_entries_distinct(G, p):
    """True if no entry (k,i,j) is listed in two groups: the groups of prox_2norm_G are then independent."""
    K = G.shape[2]
    for k in range(K):
        m = G[0, :, k] != -1
        flat = G[0, m, k] * int(p[k]) + G[1, m, k]
        if len(np.unique(flat)) != len(flat):
            return False
    return True


def prox_2norm_G_vectorized(X, G, l2):
    """prox_2norm_G for a G without repeated entries (independent groups), all groups at once -- the same arithmetic per
    group as the loop above (gather, |v|_2, scale, scatter to (i,j) and (j,i)); tests/test_oracle_golden.py checks it
    against the loop on the reference's vectors.  Used by the oracle's ext_ADMM_MGL for sizes where the loop is slow."""
    K = len(X)
    L = G.shape[1]
    present = G[0] != -1                                   # (L,K)
    V = np.zeros((L, K))
    for k in range(K):
        m = present[:, k]
        V[m, k] = X[k][G[0, m, k], G[1, m, k]]
    lam = l2 * np.sqrt(present.sum(axis=1))
    a = np.maximum(np.sqrt((V ** 2).sum(axis=1)), lam)
    Z = V * ((a - lam) / a)[:, None]
    out = {k: X[k].copy() for k in range(K)}
    for k in range(K):
        m = present[:, k]
        out[k][G[0, m, k], G[1, m, k]] = Z[m, k]
        out[k][G[1, m, k], G[0, m, k]] = Z[m, k]
    return out


def ext_stopping_criterion(Omega, Omega_t_1, Theta, L, Lambda, Lambda_t_1, X0, X1, rho, p, eps_abs, eps_rel):
    """solver/ext_admm_solver.py:325-345."""
    K = len(Omega)
    dim = ((p ** 2 + p) / 2).sum()
    n2 = lambda A: np.linalg.norm(A) ** 2
    D1 = np.sqrt(sum(n2(Omega[k]) + n2(Lambda[k]) for k in range(K)))
    D2 = np.sqrt(sum(n2(Theta[k] - L[k]) + n2(Theta[k]) for k in range(K)))
    D3 = np.sqrt(sum(n2(X0[k]) + n2(X1[k]) for k in range(K)))
    e_pri = dim * eps_abs + eps_rel * max(D1, D2)
    e_dual = dim * eps_abs + eps_rel * rho * D3
    r = np.sqrt(sum(n2(Omega[k] - Theta[k] + L[k]) + n2(Lambda[k] - Theta[k]) for k in range(K)))
    s = rho * np.sqrt(sum(n2(Omega[k] - Omega_t_1[k]) + n2(Lambda[k] - Lambda_t_1[k]) for k in range(K)))
    return r, s, e_pri, e_dual


def ext_kkt_stopping_criterion(Omega, Theta, L, Lambda, X0, X1, S, G, lambda1, lambda2, latent=False, mu1=None):
    """solver/ext_admm_solver.py:347-392; X0, X1 are the UNscaled duals (rho * scaled)."""
    K = len(S)
    t = np.zeros((6, K))
    V = {}
    n = np.linalg.norm
    for k in range(K):
        D, Q = np.linalg.eigh(Omega[k] - S[k] - X0[k])
        t[0, k] = n(Omega[k] - phiplus(1, D, Q)) / (1 + n(Omega[k]))
        t[1, k] = n(Theta[k] - prox_od_1norm(Theta[k] + X0[k] - X1[k], lambda1[k])) / (1 + n(Theta[k]))
        if latent:
            D, Q = np.linalg.eigh(L[k] - X0[k])
            t[2, k] = n(L[k] - prox_rank_norm(L[k] - X0[k], mu1[k], D, Q)) / (1 + n(L[k]))
        V[k] = Lambda[k] + X1[k]
        t[4, k] = n(Omega[k] - Theta[k] + L[k]) / (1 + n(Theta[k]))
        t[5, k] = n(Lambda[k] - Theta[k]) / (1 + n(Theta[k]))
    V = prox_2norm_G(V, G, lambda2)
    for k in range(K):
        t[3, k] = n(V[k] - Lambda[k]) / (1 + n(Lambda[k]))
    return max(n(t[i]) for i in range(6))


def ext_ADMM_MGL(S, lambda1, lambda2, reg, Omega_0, G, X0=None, X1=None, tol=1e-5, rtol=1e-4,
                 stopping_criterion='boyd', rho=1., max_iter=1000, verbose=False, measure=False, latent=False,
                 mu1=None):
    """solver/ext_admm_solver.py:18-323.  S, Omega_0, X0, X1: dicts with keys 0..K-1 and (p_k,p_k) arrays.
    No rho update in this solver (the reference has none)."""
    K = len(S)
    p = np.array([S[k].shape[0] for k in range(K)], dtype=int)
    if isinstance(lambda1, float):
        lambda1 = lambda1 * np.ones(K)
    if latent:
        if isinstance(mu1, float):
            mu1 = mu1 * np.ones(K)
        assert mu1 is not None
        assert np.all(mu1 > 0)
    assert min(lambda1.min(), lambda2) > 0
    assert reg in ['GGL']
    check_G(G, p)
    assert rho > 0
    shrink = prox_2norm_G_vectorized if (G.shape[1] > 200 and _G_entries_distinct(G, p)) else prox_2norm_G
    Omega_t = {k: Omega_0[k].copy() for k in range(K)}
    Theta_t = {k: Omega_0[k].copy() for k in range(K)}
    Lambda_t = {k: Omega_0[k].copy() for k in range(K)}
    L_t = {k: np.zeros((p[k], p[k])) for k in range(K)}
    X0_t = {k: (np.zeros((p[k], p[k])) if X0 is None else X0[k].copy()) for k in range(K)}
    X1_t = {k: (np.zeros((p[k], p[k])) if X1 is None else X1[k].copy()) for k in range(K)}
    residual = np.zeros(max_iter)
    status = ''
    for iter_t in range(max_iter):
        Omega_t_1 = Omega_t
        Omega_t = {}
        for k in range(K):
            W = Theta_t[k] - L_t[k] - X0_t[k] - (1 / rho) * S[k]
            D, Q = np.linalg.eigh(W)
            Omega_t[k] = phiplus(1 / rho, D, Q)
        for k in range(K):
            V = (Omega_t[k] + L_t[k] + X0_t[k] + Lambda_t[k] - X1_t[k]) * 0.5
            Theta_t[k] = prox_od_1norm(V, lambda1[k] / (2 * rho))
        if latent:
            for k in range(K):
                C = Theta_t[k] - X0_t[k] - Omega_t[k]
                C = (C.T + C) / 2
                D, Q = np.linalg.eigh(C)
                L_t[k] = prox_rank_norm(C, mu1[k] / rho, D, Q)
        Lambda_t_1 = Lambda_t
        Lambda_t = shrink({k: Theta_t[k] + X1_t[k] for k in range(K)}, G, lambda2 / rho)
        for k in range(K):
            X0_t[k] = X0_t[k] + Omega_t[k] - Theta_t[k] + L_t[k]
            X1_t[k] = X1_t[k] + Theta_t[k] - Lambda_t[k]
        if stopping_criterion == 'boyd':
            r_t, s_t, e_pri, e_dual = ext_stopping_criterion(Omega_t, Omega_t_1, Theta_t, L_t, Lambda_t, Lambda_t_1,
                                                             X0_t, X1_t, rho, p, tol, rtol)
            residual[iter_t] = max(r_t, s_t)
            if (r_t <= e_pri) and (s_t <= e_dual):
                status = 'optimal'
                break
        else:
            eta_A = ext_kkt_stopping_criterion(Omega_t, Theta_t, L_t, Lambda_t, {k: rho * v for k, v in X0_t.items()},
                                               {k: rho * v for k, v in X1_t.items()}, S, G, lambda1, lambda2, latent,
                                               mu1)
            residual[iter_t] = eta_A
            if eta_A <= tol:
                status = 'optimal'
                break
    if status != 'optimal':
        if stopping_criterion == 'boyd':
            status = 'primal optimal' if r_t <= e_pri else ('dual optimal' if s_t <= e_dual
                                                            else 'max iterations reached')
        else:
            status = 'max iterations reached'
    sol = {'Omega': Omega_t, 'Theta': Theta_t, 'L': L_t, 'X0': X0_t, 'X1': X1_t, 'Lambda': Lambda_t}
    info = {'status': status, 'iterations': iter_t + 1}
    if measure:
        info['residual'] = residual[:iter_t + 1]
    return sol, info


def ext_ADMM_MGL_printing(*a, **k):
    """ext_ADMM_MGL plus the reference's final status line (solver/ext_admm_solver.py:288), for the shared checks."""
    sol, info = ext_ADMM_MGL(*a, **k)
    print(f"ADMM terminated after {info['iterations']} iterations with status: {info['status']}.")
    return sol, info
