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


def _solve_grid(S, lam, mu, latent, tol, rtol, max_iter):
    """All (lambda1[, mu1]) instances as ONE batch from the reference's start (Omega_0 = X_0 = identity,
    model_selection.py:595-596).  S: (p,p) shared, or (n,p,p) one covariance matrix per instance."""
    eye = np.eye(S.shape[-1])
    res = ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=tol, rtol=rtol, latent=latent, mu1=mu, max_iter=max_iter,
                         selection_stats=True)
    # per instance: the solution, and <S,Theta>, log det Theta, non-zero count computed on the GPU
    return [s for s, _ in res], [info['selection'] for _, info in res]


def _grid_tables(S, N, sols, dev, lambda_range, mu_range, latent, method, gamma, gammas, store_all, lambda1_mask):
    """AIC / eBIC / sparsity / rank tables, best point and stored estimates of one (lambda1, mu1) grid
    (model_selection.py:583-690); sols[j*M+m] solves (lambda_range[j], mu_range[m])."""
    p = S.shape[0]
    nl, nm = len(lambda_range), len(mu_range)
    MU, LAMB = np.meshgrid(mu_range, lambda_range)
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

    # instance j*nm + m solves (lambda_range[j], mu_range[m])
    if lambda1_mask is None:
        sols, dev = _solve_grid(S, np.repeat(lambda_range, nm), np.tile(mu_range, nl) if latent else None, latent, tol,
                                rtol, max_iter)
    else:
        sols, dev = [], None
        Om0 = np.eye(p)
        for j in range(nl):
            for m in range(nm):
                kw = dict(latent=True, mu1=mu_range[m]) if latent else {}
                sol, _ = ADMM_SGL(S, lambda_range[j], Om0, X_0=np.eye(p), tol=tol, rtol=rtol, verbose=False,
                                  lambda1_mask=lambda1_mask, max_iter=max_iter, **kw)
                Om0 = sol['Omega'].copy()
                sols.append(sol)
    return _grid_tables(S, N, sols, dev, lambda_range, mu_range, latent, method, gamma, gammas, store_all, lambda1_mask)


def K_single_grid(S, lambda_range, N, method='eBIC', gamma=0.3, latent=False, mu_range=None, thresholding=False,
                  use_block=True, store_all=True, tol=1e-7, rtol=1e-7, max_iter=1000, max_batch_bytes=8 << 30):
    """Model selection for K independent Single Graphical Lasso problems on a common (lambda1, mu1) grid:
    arguments and the returned ``(est_uniform, est_indv, statistics)`` as model_selection.py:300-503 (array S only).

    The reference runs ``single_grid_search`` instance after instance; here the K x L x M problems are the
    instances of as few batches as ``max_batch_bytes`` of device stacks allow (one, typically)."""
    assert method in ('AIC', 'eBIC')
    if thresholding:
        raise NotImplementedError("thresholded estimators are outside the accelerated path")
    if not isinstance(S, np.ndarray):
        raise NotImplementedError("dictionary input (instances of different dimension) is not on the batched path")
    S = np.ascontiguousarray(S, dtype=np.float64)
    K, p = S.shape[0], S.shape[1]
    assert len(N) == K, f"N must be given as array, is given as {N}."
    lambda_range = np.atleast_1d(np.asarray(lambda_range, dtype=np.float64))
    if latent:
        assert mu_range is not None
        mu_range = np.atleast_1d(np.asarray(mu_range, dtype=np.float64))
    else:
        mu_range = np.array([0])
    nl, nm = len(lambda_range), len(mu_range)
    gammas = sorted(set(DEFAULT_GAMMAS) | {gamma})
    MU, LAMB = np.meshgrid(mu_range, lambda_range)

    # instances in the order (k, j, m); ~14 (K',p,p) stacks live on the device per batch
    per_k = nl * nm
    k_per_batch = max(1, int(max_batch_bytes // (14 * per_k * p * p * 8)))
    sols, dev = [], []
    for k0 in range(0, K, k_per_batch):
        ks = range(k0, min(K, k0 + k_per_batch))
        S_inst = np.repeat(S[list(ks)], per_k, axis=0)
        lam = np.tile(np.repeat(lambda_range, nm), len(ks))
        mu = np.tile(np.tile(mu_range, nl), len(ks)) if latent else None
        s_, d_ = _solve_grid(S_inst, lam, mu, latent, tol, rtol, max_iter)
        sols += s_
        dev += d_

    BIC = {g: np.full((K, nl, nm), np.nan) for g in gammas}
    AIC = np.full((K, nl, nm), np.nan)
    SP = np.full((K, nl, nm), np.nan)
    RANK = np.zeros((K, nl, nm))
    estimates, lowrank = dict(), dict()
    indv_T, indv_L = [], []
    for k in range(K):
        sl = slice(k * per_k, (k + 1) * per_k)
        best, est_k, lr_k, st = _grid_tables(S[k], N[k], sols[sl], dev[sl], lambda_range, mu_range, latent, method, gamma,
                                             gammas, store_all, None)
        indv_T.append(best['Theta'])
        if latent:
            indv_L.append(best['L'])
        if store_all:
            estimates[k], lowrank[k] = est_k, lr_k
        for g in gammas:
            BIC[g][k] = st['BIC'][g]
        AIC[k], SP[k], RANK[k] = st['AIC'], st['SP'], st['RANK']

    # for each lambda1 the best mu1 per instance, then the best lambda1 uniformly and individually (:443-466)
    table = AIC if method == 'AIC' else BIC[gamma]
    ix_mu = np.zeros((K, nl), dtype=int)
    score = np.full((K, nl), np.nan)
    for k in range(K):
        for j in range(nl):
            ix_mu[k, j] = np.nanargmin(table[k, j, :])
            score[k, j] = table[k, j, ix_mu[k, j]]
    score[score == -np.inf] = np.nan
    ix_uniform = np.nanargmin(score.sum(axis=0))
    ix_indv = np.nanargmin(score, axis=1)

    est_indv = {'Theta': np.stack(indv_T)}
    if latent:
        est_indv['L'] = np.stack(indv_L)
    est_uniform = None
    if store_all:
        est_uniform = {'Theta': np.stack([estimates[k][ix_uniform, ix_mu[k, ix_uniform]] for k in range(K)])}
        if latent:
            est_uniform['L'] = np.stack([lowrank[k][ix_uniform, ix_mu[k, ix_uniform]] for k in range(K)])
    statistics = {'BIC': BIC[gamma], 'AIC': AIC, 'SP': SP, 'RANK': RANK, 'LAMB': LAMB, 'MU': MU,
                  'ix_uniform': ix_uniform, 'ix_indv': ix_indv, 'ix_mu': ix_mu}
    return est_uniform, est_indv, statistics
