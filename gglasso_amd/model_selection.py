"""Model selection for the Single Graphical Lasso with the WHOLE grid solved as one batch on the GPU.

The reference's ``single_grid_search`` (helper/model_selection.py:505-692) walks the (lambda1, mu1) grid
sequentially, one ``ADMM_SGL``/``block_SGL`` call per grid point with a warm start from the previous point
(:619-633).  Grid points are independent problems on the same S, so here all L*M of them advance together
as the instances of one (L*M, p, p) stack (``gglasso_amd.batch.ADMM_SGL_batch``: one batched Omega-step and one
elementwise Theta-step per iteration, every instance with its own rho and stopping decision).  The optimum of
a grid point does not depend on its start, so the selection statistics agree with the sequential walk to the
solver tolerance; the bookkeeping below (AIC / eBIC tables, sparsity, rank, best point) follows the
reference's definitions (:769-894, helper/utils.py:17-23) and its return layout.

``grid_search`` for the multiple-graph solvers needs nothing from this module: the reference's own driver takes
``gglasso_amd.ADMM_MGL`` as its ``solver`` argument unchanged (tests/test_reference_dropin.py).
"""
import numpy as np

from .batch import ADMM_SGL_batch
from .solver import ADMM_SGL

DEFAULT_GAMMAS = (0.1, 0.3, 0.5, 0.7)        # model_selection.py:17
_NUMBER = (int, float, np.integer, np.floating)


def robust_logdet(A, t=1e-12):
    """log det A, or -inf when the smallest eigenvalue is not above ``t`` (model_selection.py:884-894)."""
    d = np.linalg.eigvalsh(A)
    if d.min() <= t:
        return -np.inf
    sign, val = np.linalg.slogdet(A)
    return sign * val


def sparsity(A):
    """ratio of non-zero off-diagonal entries (helper/utils.py:17-23)."""
    assert A.ndim == 2
    p = A.shape[0]
    return (np.count_nonzero(A) - p) / (p ** 2 - p)


def _edges(Theta, lambda1_mask=None):
    p = Theta.shape[0]
    if lambda1_mask is None:
        return (np.count_nonzero(Theta) - p) / 2          # non-zeros above the diagonal
    assert lambda1_mask.shape == Theta.shape
    E = (Theta != 0) * lambda1_mask                        # weighted by the mask, diagonal not counted
    np.fill_diagonal(E, 0)
    return E.sum() / 2


def aic_single(S, Theta, N):
    """AIC after Danaher et al. (model_selection.py:812-820)."""
    assert isinstance(N, _NUMBER)
    return N * np.sum(S * Theta) - N * robust_logdet(Theta) + _edges(Theta)


def ebic_single(S, Theta, N, gamma, lambda1_mask=None):
    """extended BIC after Drton et al. (model_selection.py:840-856)."""
    assert isinstance(N, _NUMBER)
    p = S.shape[0]
    E = _edges(Theta, lambda1_mask)
    return N * np.sum(S * Theta) - N * robust_logdet(Theta) + E * (np.log(N) + 4 * np.log(p) * gamma)


def aic(S, Theta, N):
    """(p,p) or (K,p,p) stacks (model_selection.py:769-810)."""
    if S.ndim == 2:
        return aic_single(S, Theta, N)
    Nk = np.ones(S.shape[0]) * N if isinstance(N, _NUMBER) else N
    return sum(aic_single(S[k], Theta[k], Nk[k]) for k in range(S.shape[0]))


def ebic(S, Theta, N, gamma=0.5):
    """(p,p) or (K,p,p) stacks (model_selection.py:824-866)."""
    if S.ndim == 2:
        return ebic_single(S, Theta, N, gamma)
    Nk = np.ones(S.shape[0]) * N if isinstance(N, _NUMBER) else N
    return sum(ebic_single(S[k], Theta[k], Nk[k], gamma) for k in range(S.shape[0]))


def single_grid_search(S, lambda_range, N, method='eBIC', gamma=0.3, latent=False, mu_range=None,
                       thresholding=False, use_block=True, store_all=True, tol=1e-7, rtol=1e-7, lambda1_mask=None,
                       max_iter=1000):
    """Grid search over lambda1 (and mu1 when ``latent``) for the SGL problem with selection by eBIC or AIC:
    arguments and the returned ``(best_sol, estimates, lowrank, stats)`` as model_selection.py:505-692.

    All grid points are solved as ONE batch from the reference's start (Omega_0 = X_0 = identity, :595-596);
    ``use_block`` is accepted and ignored (block splitting changes how a point is solved, not its optimum).
    ``lambda1_mask`` grids run point by point with the reference's warm start (the mask is a per-problem
    array).  ``thresholding`` (tune_threshold, :698-766) is not built."""
    assert method in ('AIC', 'eBIC')
    if thresholding:
        raise NotImplementedError("thresholded estimators are outside the accelerated path")
    S = np.ascontiguousarray(S, dtype=np.float64)
    p = S.shape[0]
    lambda_range = np.atleast_1d(np.asarray(lambda_range, dtype=np.float64))
    if latent:
        assert mu_range is not None
        mu_range = np.atleast_1d(np.asarray(mu_range, dtype=np.float64))
    else:
        mu_range = np.array([0])
    nl, nm = len(lambda_range), len(mu_range)
    gammas = sorted(set(DEFAULT_GAMMAS) | {gamma})
    MU, LAMB = np.meshgrid(mu_range, lambda_range)

    # instance j*nm + m solves (lambda_range[j], mu_range[m])
    if lambda1_mask is None:
        lam = np.repeat(lambda_range, nm)
        mu = np.tile(mu_range, nl) if latent else None
        eye = np.eye(p)
        res = ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=tol, rtol=rtol, latent=latent, mu1=mu,
                             max_iter=max_iter, selection_stats=True)
        sols = [s for s, _ in res]
        dev = [info['selection'] for _, info in res]      # <S,Theta>, log det, non-zero counts: computed on the GPU
    else:
        sols = []
        Om0 = np.eye(p)
        for j in range(nl):
            for m in range(nm):
                kw = dict(latent=True, mu1=mu_range[m]) if latent else {}
                sol, _ = ADMM_SGL(S, lambda_range[j], Om0, X_0=np.eye(p), tol=tol, rtol=rtol, verbose=False,
                                  lambda1_mask=lambda1_mask, max_iter=max_iter, **kw)
                Om0 = sol['Omega'].copy()
                sols.append(sol)
        dev = None

    BIC = {g: np.full((nl, nm), np.nan) for g in gammas}
    AIC = np.full((nl, nm), np.nan)
    SP = np.full((nl, nm), np.nan)
    RANK = np.zeros((nl, nm))
    estimates = np.zeros((nl, nm, p, p)) if store_all else None
    lowrank = np.zeros((nl, nm, p, p)) if store_all else None
    best_sol, curr_min = dict(), np.inf
    for j in range(nl):
        for m in range(nm):
            sol = sols[j * nm + m]
            Theta = sol['Theta']
            if latent:
                if store_all:
                    lowrank[j, m] = sol['L']
                # on the host: matrix_rank's tolerance p*eps*|L| (:638) is below what the device eigensolvers resolve
                RANK[j, m] = np.linalg.matrix_rank(sol['L'], hermitian=True)
            # the criteria share the expensive terms <S,Theta> and log det Theta: from the
            # device statistics of the batch, or on the host for the point-by-point (mask) walk
            if dev is not None:
                d = dev[j * nm + m]
                fit = N * d['Sdot'] - N * d['logdet']
                E0 = (d['nnz'] - p) / 2
                E = E0
                SP[j, m] = (d['nnz'] - p) / (p ** 2 - p)
            else:
                fit = N * np.sum(S * Theta) - N * robust_logdet(Theta)
                E0, E = _edges(Theta), _edges(Theta, lambda1_mask)
                SP[j, m] = sparsity(Theta)
            AIC[j, m] = fit + E0
            for g in gammas:
                BIC[g][j, m] = fit + E * (np.log(N) + 4 * np.log(p) * g)
            if store_all:
                estimates[j, m] = Theta
            score = BIC[gamma][j, m] if method == 'eBIC' else AIC[j, m]
            if score < curr_min:
                curr_min = score
                best_sol = dict(sol)
    AIC[AIC == -np.inf] = np.nan
    for g in gammas:
        BIC[g][BIC[g] == -np.inf] = np.nan
    table = AIC if method == 'AIC' else BIC[gamma]
    ix = np.unravel_index(np.nanargmin(table), table.shape)
    stats = {'BIC': BIC, 'AIC': AIC, 'SP': SP, 'RANK': RANK, 'LAMBDA': LAMB, 'MU': MU, 'TAU': None,
             'BEST': {'lambda1': LAMB[ix], 'mu1': MU[ix]}, 'GAMMA': gammas}
    return best_sol, estimates, lowrank, stats
