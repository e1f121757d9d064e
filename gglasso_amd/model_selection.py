"""Model selection for the Single Graphical Lasso with the WHOLE grid solved as one batch on the GPU.

The reference's ``single_grid_search`` (helper/model_selection.py:505-692) walks the (lambda1, mu1) grid
sequentially, one ``ADMM_SGL``/``block_SGL`` call per grid point with a warm start from the previous point
(:619-633).  Grid points are independent problems on the same S, so here all L*M of them advance together
as the instances of one (L*M, p, p) stack (``gglasso_amd.batch.ADMM_SGL_batch``: one batched Omega-step and one
elementwise Theta-step per iteration, every instance with its own rho and stopping decision).  The optimum of
a grid point does not depend on its start, so the selection statistics agree with the sequential walk to the
solver tolerance; the bookkeeping below (AIC / eBIC tables, sparsity, rank, best point) follows the
reference's definitions (:769-894, helper/utils.py:17-23) and its return layout.

``grid_search`` (lambda1 x lambda2 grid of the multiple-graph problems, :55-298) is at the end of this module: the whole
grid as one batch on the GPU (``gglasso_amd.batch.ADMM_MGL_batch``), or -- for any other solver callable -- the
reference's sequential warm-started walk.  The reference's own ``grid_search`` also takes ``gglasso_amd.ADMM_MGL``
as its ``solver`` argument unchanged (tests/test_reference_dropin.py).
"""
import numpy as np

from .batch import ADMM_SGL_batch
from .solver import ADMM_SGL, latent_rank

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


def _solve_grid(S, lam, mu, latent, tol, rtol, max_iter, tau_range=None, fetch=None, select=None):
    """All (lambda1[, mu1]) instances as ONE batch from the reference's start (Omega_0 = X_0 = identity,
    model_selection.py:595-596).  S: (p,p) shared, or (n,p,p) one covariance matrix per instance.  ``tau_range``: also
    score every instance's estimate thresholded at every tau on the GPU (tune_threshold, :707-737)."""
    eye = np.eye(S.shape[-1])
    # fetch / select: a grid walk reads Theta (and L) of every point; only the point it selects is returned whole, so only
    # that point's Omega and X are downloaded (ADMM_SGL_batch)
    res = ADMM_SGL_batch(S, lam, Omega_0=eye, X_0=eye, tol=tol, rtol=rtol, latent=latent, mu1=mu, max_iter=max_iter,
                         selection_stats=True, tau_range=tau_range, fetch=fetch, select=select)
    # per instance: the solution, and <S,Theta>, log det Theta, non-zero count computed on the GPU
    return [s for s, _ in res], [info['selection'] for _, info in res]


def _grid_tables(S, N, sols, dev, lambda_range, mu_range, latent, method, gamma, gammas, store_all, lambda1_mask,
                 thresholding=False, return_index=False):
    """AIC / eBIC / sparsity / rank tables, best point and stored estimates of one (lambda1, mu1) grid
    (model_selection.py:583-690); sols[j*M+m] solves (lambda_range[j], mu_range[m]).  thresholding: every point's
    Theta is replaced by its best thresholded version (tune_threshold, :641-650) before the criteria are taken."""
    p = S.shape[0]
    nl, nm = len(lambda_range), len(mu_range)
    MU, LAMB = np.meshgrid(mu_range, lambda_range)
    BIC = {g: np.full((nl, nm), np.nan) for g in gammas}
    AIC = np.full((nl, nm), np.nan)
    SP = np.full((nl, nm), np.nan)
    RANK = np.zeros((nl, nm))
    TAU = np.zeros((nl, nm)) if thresholding else None
    estimates = np.zeros((nl, nm, p, p)) if store_all else None
    lowrank = np.zeros((nl, nm, p, p)) if store_all else None
    best_sol, curr_min, best_k = dict(), np.inf, None
    for j in range(nl):
        for m in range(nm):
            sol = sols[j * nm + m]
            Theta = sol['Theta']
            if latent:
                if store_all:
                    lowrank[j, m] = sol['L']
            d = dev[j * nm + m] if dev is not None else None
            if latent:
                # matrix_rank(L) of :638: from the batch (the device's eigenvalue count where it rebuilt L, numpy's rule
                # otherwise), or numpy's rule on the host for the point-by-point walk
                RANK[j, m] = d['rank'] if (d is not None and 'rank' in d) else latent_rank(sol['L'])
            if thresholding and d is not None and 'threshold' in d:
                # scored on the GPU for every tau of the default range; the statistics of the chosen one become the point's
                jt = _pick_threshold(d['threshold'], N, p, method, gamma)
                TAU[j, m] = default_tau_range()[jt]
                sol['Theta'] = Theta = _apply_threshold(Theta, TAU[j, m])
                d = dict(zip(('Sdot', 'logdet', 'nnz'), d['threshold'][jt, :3]))
            elif thresholding:
                sol['Theta'], TAU[j, m], _ = tune_threshold(Theta, S, N, tau_range=None, method=method, gamma=gamma)
                Theta = sol['Theta']
                d = None
            # the criteria share the expensive terms <S,Theta> and log det Theta: from the
            # device statistics of the batch, or on the host for the point-by-point (mask) walk
            if d is not None:
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
                best_k = j * nm + m
    AIC[AIC == -np.inf] = np.nan
    for g in gammas:
        BIC[g][BIC[g] == -np.inf] = np.nan
    table = AIC if method == 'AIC' else BIC[gamma]
    ix = np.unravel_index(np.nanargmin(table), table.shape)
    stats = {'BIC': BIC, 'AIC': AIC, 'SP': SP, 'RANK': RANK, 'LAMBDA': LAMB, 'MU': MU, 'TAU': TAU,
             'BEST': {'lambda1': LAMB[ix], 'mu1': MU[ix]}, 'GAMMA': gammas}
    if return_index:
        return best_sol, estimates, lowrank, stats, best_k
    return best_sol, estimates, lowrank, stats


def single_grid_search(S, lambda_range, N, method='eBIC', gamma=0.3, latent=False, mu_range=None,
                       thresholding=False, use_block=True, store_all=True, tol=1e-7, rtol=1e-7, lambda1_mask=None,
                       max_iter=1000):
    """Grid search over lambda1 (and mu1 when ``latent``) for the SGL problem with selection by eBIC or AIC:
    arguments and the returned ``(best_sol, estimates, lowrank, stats)`` as model_selection.py:505-692.

    All grid points are solved as ONE batch from the reference's start (Omega_0 = X_0 = identity, :595-596);
    ``use_block`` is accepted and ignored (block splitting changes how a point is solved, not its optimum).
    ``lambda1_mask`` grids run point by point with the reference's warm start (the mask is a per-problem
    array).  ``thresholding``: each point's estimate is replaced by its best thresholded version (tune_threshold,
    :698-737) before the criteria are taken, as in the reference; the 20 candidate thresholds of every grid point are scored
    on the GPU (``ggl_threshold_scan``), the mask walk scores them on the host."""
    assert method in ('AIC', 'eBIC')
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
        # the tables are taken while the batch's device state is still there (``select``): Theta (and L) of all points come
        # down, Omega and X of the selected one only -- what the reference returns as best_sol (:652-660)
        out = {}

        def select(results):
            tabs = _grid_tables(S, N, [sol for sol, _ in results], [info['selection'] for _, info in results], lambda_range,
                                mu_range, latent, method, gamma, gammas, store_all, None, thresholding, return_index=True)
            out['tables'] = tabs[:4]
            return tabs[4], tabs[0]

        _solve_grid(S, np.repeat(lambda_range, nm), np.tile(mu_range, nl) if latent else None, latent, tol, rtol, max_iter,
                    default_tau_range() if thresholding else None, fetch=('Theta', 'L'), select=select)
        return out['tables']
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
    return _grid_tables(S, N, sols, dev, lambda_range, mu_range, latent, method, gamma, gammas, store_all, lambda1_mask,
                        thresholding)


def _K_single_grid_dict(S, lambda_range, N, method, gamma, latent, mu_range, thresholding, use_block, store_all, tol,
                        rtol, max_iter):
    """K_single_grid for a dict of instances of DIFFERENT dimension (model_selection.py:300-503 with dict S): every
    instance's (lambda1, mu1) grid is its own batch (``single_grid_search``); the selection across instances follows
    the reference (:443-466)."""
    K = len(S.keys())
    assert len(N) == K, f"N must be given as array, is given as {N}."
    lambda_range = np.atleast_1d(np.asarray(lambda_range, dtype=np.float64))
    mu_r = np.atleast_1d(np.asarray(mu_range, dtype=np.float64)) if latent else np.array([0])
    nl, nm = len(lambda_range), len(mu_r)
    MU, LAMB = np.meshgrid(mu_r, lambda_range)
    BIC, AIC = np.full((K, nl, nm), np.nan), np.full((K, nl, nm), np.nan)
    SP, RANK = np.full((K, nl, nm), np.nan), np.zeros((K, nl, nm))
    estimates, lowrank, indv_T, indv_L = dict(), dict(), dict(), dict()
    for k in range(K):
        best, estimates[k], lowrank[k], st = single_grid_search(S[k], lambda_range, N[k], method, gamma, latent, mu_range,
                                                                thresholding, use_block, True, tol, rtol, None, max_iter)
        indv_T[k] = best['Theta']
        if latent:
            indv_L[k] = best['L']
        BIC[k], AIC[k], SP[k], RANK[k] = st['BIC'][gamma], st['AIC'], st['SP'], st['RANK']
    table = AIC if method == 'AIC' else BIC
    ix_mu = np.zeros((K, nl), dtype=int)
    score = np.full((K, nl), np.nan)
    for k in range(K):
        for j in range(nl):
            ix_mu[k, j] = np.nanargmin(table[k, j, :])
            score[k, j] = table[k, j, ix_mu[k, j]]
    score[score == -np.inf] = np.nan
    ix_uniform = np.nanargmin(score.sum(axis=0))
    ix_indv = np.nanargmin(score, axis=1)
    est_indv = {'Theta': indv_T}
    est_uniform = {'Theta': {k: estimates[k][ix_uniform, ix_mu[k, ix_uniform]] for k in range(K)}}
    if latent:
        est_indv['L'] = indv_L
        est_uniform['L'] = {k: lowrank[k][ix_uniform, ix_mu[k, ix_uniform]] for k in range(K)}
    statistics = {'BIC': BIC, 'AIC': AIC, 'SP': SP, 'RANK': RANK, 'LAMB': LAMB, 'MU': MU, 'ix_uniform': ix_uniform,
                  'ix_indv': ix_indv, 'ix_mu': ix_mu}
    return est_uniform, est_indv, statistics


def K_single_grid(S, lambda_range, N, method='eBIC', gamma=0.3, latent=False, mu_range=None, thresholding=False,
                  use_block=True, store_all=True, tol=1e-7, rtol=1e-7, max_iter=1000, max_batch_bytes=8 << 30):
    """Model selection for K independent Single Graphical Lasso problems on a common (lambda1, mu1) grid:
    arguments and the returned ``(est_uniform, est_indv, statistics)`` as model_selection.py:300-503 (array S only).

    The reference runs ``single_grid_search`` instance after instance; here the K x L x M problems are the
    instances of as few batches as ``max_batch_bytes`` of device stacks allow (one, typically)."""
    assert method in ('AIC', 'eBIC')
    if isinstance(S, dict):
        return _K_single_grid_dict(S, lambda_range, N, method, gamma, latent, mu_range, thresholding, use_block, store_all,
                                   tol, rtol, max_iter)
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
        s_, d_ = _solve_grid(S_inst, lam, mu, latent, tol, rtol, max_iter, default_tau_range() if thresholding else None,
                             fetch=('Theta', 'L'))            # (K_single_grid uses Theta and L of the points, nothing else)
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
                                             gammas, store_all, None, thresholding)
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


# -----------------------------------------------------------------------------------------------------------------
# lambda1 x lambda2 grid for the multiple-graph problems
# -----------------------------------------------------------------------------------------------------------------
N_TAU = 20      # model_selection.py:18


def lambda_parametrizer(l1=0.05, w2=0.5):
    """model_selection.py:20-25."""
    a = 1 / np.sqrt(2)
    return (w2 * l1) / (a * (1 - w2))


def lambda_grid(l1, l2=None, w2=None):
    """model_selection.py:32-52: lambda1 changes over the columns, lambda2 over the rows.  (The reference squeezes
    the meshgrid, which breaks its own MAIN LOOP for one-row or one-column grids; the grids stay 2-D here.)"""
    assert np.all(l2 is not None) | np.all(w2 is not None), \
        "Either a range of lambda2 or w2 values have to be specified"
    if np.all(l2 is not None):
        L1, L2 = np.meshgrid(l1, l2)
    else:
        l1grid, w2grid = np.meshgrid(l1, w2)
        L2 = lambda_parametrizer(l1grid, w2grid)
        L1 = l1grid.copy()
    return np.atleast_2d(L1), np.atleast_2d(L2)


def mean_sparsity(Theta):
    """helper/utils.py:25-31."""
    if isinstance(Theta, dict):
        return np.mean([sparsity(Theta[k]) for k in Theta.keys()])
    return np.mean([sparsity(Theta[k]) for k in range(Theta.shape[0])])


def thresholding(A, tau):
    """model_selection.py:698-705: entries with |a| <= tau are set to zero, the diagonal is kept."""
    mask = (np.abs(A) > tau)
    np.fill_diagonal(mask, 1.)
    return A * mask


_apply_threshold = thresholding        # the functions above take a flag of that name


def default_tau_range():
    return np.logspace(-12, -1, N_TAU)             # model_selection.py:714


def _pick_threshold(table, N, p, method, gamma):
    """tune_threshold's choice (model_selection.py:718-735) from the (ntau, 4) device table of
    {<S,T>, log det T, count_nonzero(T), lambda_min(T)}: index of the best tau (first of equals, nan = not definite)."""
    E = (table[:, 2] - p) / 2
    scores = N * table[:, 0] - N * table[:, 1] + E * ((np.log(N) + 4 * np.log(p) * gamma) if method == 'eBIC' else 1.0)
    scores[scores == np.inf] = np.nan
    return int(np.nanargmin(scores))


def tune_threshold(Theta, S, N, tau_range=None, method='eBIC', gamma=0.1):
    """model_selection.py:707-737."""
    if tau_range is None:
        tau_range = default_tau_range()
    assert np.all(tau_range > 0)
    scores = np.zeros(len(tau_range))
    for j in range(len(tau_range)):
        T = thresholding(Theta, tau_range[j])
        scores[j] = ebic_single(S, T, N, gamma) if method == 'eBIC' else aic_single(S, T, N)
    scores[scores == np.inf] = np.nan
    opt_tau = tau_range[np.nanargmin(scores)]
    return thresholding(Theta, opt_tau), opt_tau, scores


def tune_multiple_threshold(Theta, S, N, tau_range, method='eBIC', gamma=0.1):
    """model_selection.py:739-765 (arrays or dicts)."""
    K = len(S.keys()) if isinstance(S, dict) else S.shape[0]
    t_Theta = Theta.copy()
    score = dict()
    tau = np.zeros(K)
    for k in range(K):
        t_Theta[k], tau[k], score[k] = tune_threshold(Theta[k], S[k], N[k], tau_range, method, gamma)
    return t_Theta, tau, score


def _criteria(S, Theta, N, K):
    """Per-instance fit term N (<S,Theta> - log det Theta) and edge count of aic_single / ebic_single (:812-856)."""
    fit = np.array([N[k] * np.sum(S[k] * Theta[k]) - N[k] * robust_logdet(Theta[k]) for k in range(K)])
    E = np.array([_edges(Theta[k]) for k in range(K)])
    return fit, E


def grid_search(solver, S, N, p, reg, l1, l2=None, w2=None, method='eBIC', gamma=0.3, G=None, latent=False,
                mu_range=None, ix_mu=None, thresholding=False, tol=1e-7, rtol=1e-7, verbose=False, group=None,
                max_batch_bytes=8 << 30, batched=None):
    """Model selection for the multiple-graph problems over a lambda1 x lambda2 grid with AIC / eBIC -- arguments and
    the returned ``(stats, ix, curr_best)`` as the reference's ``grid_search`` (helper/model_selection.py:55-298).

    With ``solver`` = ``gglasso_amd.ADMM_MGL`` and S an array, the WHOLE grid is solved as one batch on the GPU
    (``gglasso_amd.batch.ADMM_MGL_batch``: grid points are independent problems on the same S; one batched Omega-step
    over all (grid points x K) matrices and one Theta-step launch per iteration, every point with its own rho and
    stopping decision) from the identity start, and the criteria's expensive terms <S,Theta>, log det Theta,
    count_nonzero come from the device.  The reference walks the grid column by column with a warm start from the
    previous point (:208-224); the optimum of a point does not depend on its start, so tables and selection agree
    to the solver tolerance.  ``group``: a torch.distributed process group -- the grid points are dealt round-robin
    over its ranks (``gglasso_amd.dist.shard_grid``; replicas only, results gathered on every rank).
    ``solver`` = ``gglasso_amd.ext_ADMM_MGL`` with dict S and G (instances of different dimension) is batched the same way
    (``gglasso_amd.ext_solver.ext_ADMM_MGL_batch``; the criteria then come from the host, per instance dimension).
    Any other solver callable (the reference's own solvers) and ``batched=False`` take the reference's sequential
    warm-started walk with that callable.  ``thresholding`` tunes a thresholded
    estimator per grid point on the host (tune_multiple_threshold, :739-765) in either mode."""
    from .solver import ADMM_MGL
    assert method in ['AIC', 'eBIC']
    assert reg in ['FGL', 'GGL']
    if isinstance(S, dict):
        K = len(S.keys())
    elif isinstance(S, np.ndarray):
        K = S.shape[0]
    else:
        raise Exception("S must be specified either as array or dict.")
    assert len(N) == K, f"N must be given as array, is given as {N}."
    if latent:
        assert np.all(mu_range > 0)
    L1, L2 = lambda_grid(l1, l2, w2)
    if verbose:
        print("Grid of lambda1/lambda2:")
        print(L1)
        print(L2)
    grid1, grid2 = L1.shape
    gammas = sorted(set(DEFAULT_GAMMAS) | {gamma})
    AIC = np.nan * np.zeros((grid1, grid2))
    BIC = {g: np.nan * np.zeros((grid1, grid2)) for g in gammas}
    SP = np.nan * np.zeros((grid1, grid2))
    RANK = np.nan * np.zeros((K, grid1, grid2))
    TAU = np.zeros((K, grid1, grid2)) if thresholding else None
    from .ext_solver import ext_ADMM_MGL as _ext_solver
    ext_batched = (batched is None or batched) and (solver is _ext_solver) and isinstance(S, dict) and reg == 'GGL' \
        and G is not None and group is None
    if ext_batched:
        batched = False
    if batched is None:
        # both penalties fall back to the sequential walk beyond what their batched Theta kernel takes (the library
        # exports the limits; ADVICE r2: FGL used to surface the limit as an AssertionError instead)
        from ._lib import theta_limits
        batched = (solver is ADMM_MGL) and isinstance(S, np.ndarray) and K <= theta_limits()[reg]
    order = [(g1, g2) for g2 in range(grid2) for g1 in range(grid1)]       # down the columns, as the reference

    sols = {}
    dev, dev_thr, dev_rank = {}, {}, {}
    if batched:
        from .batch import ADMM_MGL_batch
        pdim = S.shape[1]
        mine = list(range(len(order)))
        if group is not None:
            import torch.distributed as dist
            from .dist import shard_grid
            mine = shard_grid(len(order), dist.get_world_size(group), dist.get_rank(group))
        per_batch = max(1, int(max_batch_bytes // (14 * K * pdim * pdim * 8)))
        local = []
        for b0 in range(0, len(mine), per_batch):
            idx = mine[b0:b0 + per_batch]
            lam1 = np.array([L1[order[i]] for i in idx])
            lam2 = np.array([L2[order[i]] for i in idx])
            mu = np.stack([mu_range[ix_mu[:, order[i][1]]] for i in idx]) if latent else None
            # without thresholding the walk below reads Theta and L of the points and returns the whole solution of the best one
            # only: Theta and L of all points come down, Omega and X of each batch's best (the overall best is one of them)
            sel = None
            if not thresholding:
                pS_, Nk_ = S.shape[1], np.asarray(N, dtype=np.float64) * np.ones(K)

                def sel(results):
                    best, best_g = np.inf, None
                    for g, (_, info) in enumerate(results):
                        d = info['selection']
                        fit, E = Nk_ * d[:, 0] - Nk_ * d[:, 1], (d[:, 2] - pS_) / 2
                        sc = np.sum(fit + E * (np.log(Nk_) + 4 * np.log(pS_) * gamma)) if method == 'eBIC' else np.sum(fit + E)
                        if sc < best:
                            best, best_g = sc, g
                    return best_g, None
            res = ADMM_MGL_batch(S, lam1, lam2, reg, tol=tol, rtol=rtol, latent=latent, mu1=mu, selection_stats=True,
                                 tau_range=default_tau_range() if thresholding else None,
                                 fetch=None if thresholding else ('Theta', 'L'), select=sel)
            local += [(i, r) for i, r in zip(idx, res)]
        if group is not None:
            gathered = [None] * dist.get_world_size(group)
            dist.all_gather_object(gathered, local, group=group)
            local = [x for part in gathered for x in part]
        for i, (sol, info) in local:
            sols[order[i]] = sol
            dev[order[i]] = info['selection']
            if thresholding:
                dev_thr[order[i]] = info['threshold']
            if latent:
                dev_rank[order[i]] = info['rank']
    elif ext_batched:
        from .ext_solver import ext_ADMM_MGL_batch
        pmax = max(S[k].shape[0] for k in range(K))
        per_batch = max(1, int(max_batch_bytes // (17 * K * pmax * pmax * 8)))
        for b0 in range(0, len(order), per_batch):
            idx = order[b0:b0 + per_batch]
            lam1 = [L1[g] for g in idx]
            lam2 = np.array([L2[g] for g in idx])
            mu = np.stack([mu_range[ix_mu[:, g[1]]] for g in idx]) if latent else None
            res = ext_ADMM_MGL_batch(S, lam1, lam2, reg, G, tol=tol, rtol=rtol, latent=latent, mu1=mu)
            for g, (sol, info) in zip(idx, res):
                sols[g] = sol
    else:
        kwargs = {'reg': reg, 'S': S, 'tol': tol, 'rtol': rtol, 'verbose': False, 'measure': False}
        if isinstance(S, dict):
            kwargs['Omega_0'] = {k: np.eye(S[k].shape[0]) for k in range(K)}      # id_dict, ext_admm_helper.py:9-16
            kwargs['G'] = G
        else:
            kwargs['Omega_0'] = np.stack([np.eye(S.shape[1])] * K)               # id_array, utils.py:10-15
        for (g1, g2) in order:
            kwargs['lambda1'], kwargs['lambda2'] = L1[g1, g2], L2[g1, g2]
            if latent:
                kwargs['latent'] = True
                kwargs['mu1'] = mu_range[ix_mu[:, g2]].copy()
            sol, info = solver(**kwargs)
            kwargs['Omega_0'] = sol['Omega'].copy()                               # warm start (:224)
            sols[(g1, g2)] = sol

    curr_min, curr_best = np.inf, None
    no_thr_min, no_thr_params, no_thr_best = np.inf, None, None
    for (g1, g2) in order:
        sol = sols[(g1, g2)]
        d = dev.get((g1, g2))
        if d is not None:
            # the batch's device statistics; with thresholding also those of every instance's estimate at every tau
            pS = S.shape[1]
            fit = N * d[:, 0] - N * d[:, 1]
            nnz = d[:, 2].copy()
            E = (nnz - pS) / 2
            if thresholding:
                score0 = np.sum(fit + E * (np.log(N) + 4 * np.log(pS) * gamma))
                if score0 < no_thr_min:
                    no_thr_min, no_thr_best = score0, sol.copy()
                    no_thr_params = {'lambda1': L1[g1, g2], 'lambda2': L2[g1, g2]}
                sol['Theta'] = sol['Theta'].copy()
                for k in range(K):
                    tab = dev_thr[(g1, g2)][k]
                    jt = _pick_threshold(tab, N[k], pS, method, gamma)
                    TAU[k, g1, g2] = default_tau_range()[jt]
                    sol['Theta'][k] = _apply_threshold(sol['Theta'][k], TAU[k, g1, g2])
                    fit[k], nnz[k] = N[k] * tab[jt, 0] - N[k] * tab[jt, 1], tab[jt, 2]
                E = (nnz - pS) / 2
            SP[g1, g2] = np.mean((nnz - pS) / (pS ** 2 - pS))
        else:
            if thresholding:
                fit0, E0 = _criteria(S, sol['Theta'], N, K)
                pk = np.array([S[k].shape[0] for k in range(K)])
                score0 = np.sum(fit0 + E0 * (np.log(N) + 4 * np.log(pk) * gamma))
                if score0 < no_thr_min:
                    no_thr_min, no_thr_best = score0, sol.copy()
                    no_thr_params = {'lambda1': L1[g1, g2], 'lambda2': L2[g1, g2]}
                sol['Theta'], TAU[:, g1, g2], _ = tune_multiple_threshold(sol['Theta'], S, N, tau_range=None,
                                                                          method=method, gamma=gamma)
            fit, E = _criteria(S, sol['Theta'], N, K)
            SP[g1, g2] = mean_sparsity(sol['Theta'])
        pk = np.array([S[k].shape[0] for k in range(K)])
        Nk = np.asarray(N, dtype=np.float64)
        AIC[g1, g2] = np.sum(fit + E)
        for g in gammas:
            BIC[g][g1, g2] = np.sum(fit + E * (np.log(Nk) + 4 * np.log(pk) * g))
        if latent:
            if (g1, g2) in dev_rank:
                RANK[:, g1, g2] = dev_rank[(g1, g2)]
            else:
                RANK[:, g1, g2] = [latent_rank(sol['L'][k]) for k in range(K)]
        score = BIC[gamma][g1, g2] if method == 'eBIC' else AIC[g1, g2]
        if score < curr_min:
            curr_min = score
            curr_best = sol.copy()
        if verbose:
            print(f"Grid point: (l1,l2): {(L1[g1, g2], L2[g1, g2])}, sparsity: {np.round(SP[g1, g2], 3)}, "
                  f"best score: {np.round(curr_min, 1)}")
    if method == 'AIC':
        AIC[AIC == -np.inf] = np.nan
        ix = np.unravel_index(np.nanargmin(AIC), AIC.shape)
    else:
        for g in gammas:
            BIC[g][BIC[g] == -np.inf] = np.nan
        ix = np.unravel_index(np.nanargmin(BIC[gamma]), BIC[gamma].shape)
    if verbose:
        print(f"Best regularization parameters: (l1,l2): {(L1[ix], L2[ix])}")
    stats = {'BIC': BIC, 'AIC': AIC, 'SP': SP, 'RANK': RANK, 'TAU': TAU, 'L1': L1, 'L2': L2,
             'BEST': {'lambda1': L1[ix], 'lambda2': L2[ix]}, 'GAMMA': gammas}
    if thresholding:
        stats['NO_THRESHOLDING_SOL'] = no_thr_best
        stats['NO_THRESHOLDING_BEST'] = no_thr_params
    return stats, ix, curr_best
