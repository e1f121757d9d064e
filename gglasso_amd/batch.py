"""Batched lambda path for the Single Graphical Lasso: K independent ``ADMM_SGL`` problems on the same
S advance together on one GPU -- one batched Omega-step (K matrices per launch instead of one) and one
elementwise Theta-step per iteration.

This is the regime of the reference's ``single_grid_search`` (helper/model_selection.py:505-692), which
solves the lambda1 grid sequentially (each solve is the loop of solver/single_admm_solver.py:157-214).
Every instance keeps its OWN rho, residuals, rho updates and stopping decision, exactly as if
``ADMM_SGL(S, lambda1[k], Omega_0, ...)`` had been called on its own, so each returned (sol, info) equals
the independent solve (the reference's sequential warm start, model_selection.py:632-633, only changes
iteration counts, not the optimum).  An instance's solution is snapshotted at the iteration it converges;
it keeps iterating harmlessly until the whole batch is done.
"""
import numpy as np

from . import solver as _solver
from .solver import as_c


def ADMM_SGL_batch(S, lambda1, Omega_0=None, Theta_0=None, X_0=None, rho=1., max_iter=1000, tol=1e-7,
                   rtol=1e-4, update_rho=True, verbose=False, latent=False, mu1=None, lambda1_mask=None,
                   selection_stats=False, dims=None, tau_range=None, compact=True, fetch=None, select=None):
    """Solve ``ADMM_SGL(S, lambda1[k], ...)`` for every k of the 1-D array ``lambda1`` at once.

    S: (p,p) shared by all instances, or (K,p,p) with one covariance matrix per instance (what
    ``block_SGL`` needs for equally sized blocks).  Omega_0 / Theta_0 / X_0: (p,p) shared start or (K,p,p) per instance (default identity /
    Omega_0 / zeros, as in single_admm_solver.py:129-137).  mu1: scalar or (K,) when ``latent``.
    Returns a list of K ``(sol, info)`` pairs with the reference's keys; ``info`` additionally carries
    ``'iterations'`` and the final ``'rho'``.  ``fetch`` (names, e.g. ``('Theta', 'L')``; needs the C loop): download only these
    stacks -- a grid walk looks at Theta (and L) of every point and returns the whole solution of ONE (the download of a
    20-point grid at p = 1000 is 480 MB otherwise); ``select(results)`` is then called before the device state goes away and
    returns ``(k, target)``: point k's missing stacks are fetched into the dict ``target`` (and into its own sol).  ``selection_stats``: also keep a device snapshot of every instance's
    solution's Theta and attach ``info['selection'] = {'Sdot','logdet','nnz','lambda_min'}`` computed on the GPU
    (what the AIC / eBIC tables of model selection are made of); with ``latent`` also ``'rank'``:
    numpy.linalg.matrix_rank of the instance's L (an int; every returned L is rebuilt from one eigendecomposition of
    its last L-step's input at the end of the batch, ``HipEngine.finalize_L``, and where that happened on the device the
    count of eigenvalues above the threshold comes with it); with ``tau_range`` also ``'threshold'``: the
    (len(tau_range), 4) table of the same four statistics for the estimate thresholded at every tau (tune_threshold,
    helper/model_selection.py:707-737).
    ``dims`` (K,) ints: problems of DIFFERENT dimension in one batch -- instance k is the leading (dims[k], dims[k]) block
    of its slot and the caller has padded S / Omega_0 / Theta_0 with an identity block and X_0 with zeros behind it (a
    decoupled fixed point of the iteration; ``pad_blocks`` builds such stacks).  Residuals, ``dim`` and the stopping
    decision of an instance are those of its block alone, and its solution is returned un-padded.  ``lambda1_mask`` may
    then be (K,p,p): one mask per instance (with one lambda1 per instance).
    A point whose data are not finite (a NaN in its S) ends with ``info['status'] == 'solver error'`` and costs that point
    only; ``compact``: once a quarter of the live slots hold finished points, the rest go on in a smaller stack."""
    S = as_c(S)
    assert S.ndim in (2, 3) and S.shape[-1] == S.shape[-2]
    p = S.shape[-1]
    lam = as_c(np.atleast_1d(lambda1)).reshape(-1)
    if S.ndim == 3 and len(lam) == 1:
        lam = np.repeat(lam, S.shape[0])      # K different covariance matrices, one lambda1
    K = len(lam)
    assert S.ndim == 2 or S.shape[0] == K
    assert np.all(lam > 0), "lambda1 should be positive"
    assert rho > 0
    lam_pp = None
    if lambda1_mask is not None:
        assert lambda1_mask.shape in ((p, p), (K, p, p))
        assert np.all(lambda1_mask >= 0)
        assert np.all(np.abs(np.swapaxes(lambda1_mask, -1, -2) - lambda1_mask) <= 1e-5)
        lam_pp = as_c(lambda1_mask)          # the per-instance factor lambda1[k] multiplies it on the device
    if dims is not None:
        dims = np.asarray(dims, dtype=np.int64).reshape(-1)
        assert len(dims) == K and S.ndim == 3 and not latent
        assert np.all((dims >= 1) & (dims <= p))
    if latent:
        assert mu1 is not None
        mu = as_c(np.broadcast_to(np.asarray(mu1, dtype=np.float64), (K,)))
        assert np.all(mu > 0)
    else:
        mu = None

    def stack(A, default):
        # a shared (p,p) start stays a broadcast VIEW: the engine uploads it once and replicates it on the device
        if A is None or len(A) == 0:
            A = default
        A = np.asarray(A, dtype=np.float64)
        # (p,p), (1,p,p) [shared] or (K,p,p); anything else is an error here, not a read past the end of the buffer in C
        assert A.ndim in (2, 3) and A.shape[-2:] == (p, p) and (A.ndim == 2 or A.shape[0] in (1, K)), \
            f"start array of shape {A.shape}: expected ({p},{p}) or ({K},{p},{p})"
        if A.ndim == 3 and A.shape[0] == K and (A.strides[0] == 0 or K == 1):
            return A
        if A.ndim == 3 and A.shape[0] == K:
            return as_c(A)
        return np.broadcast_to(as_c(A if A.ndim == 2 else A[0]), (K, p, p))

    Om0 = stack(Omega_0, np.eye(p))
    Th0 = stack(Theta_0, Om0)
    X0 = stack(X_0, np.zeros((p, p)))
    eng = _solver.ENGINE(np.broadcast_to(S, (K, p, p)), Om0, Th0, X0)
    engines = [eng]                  # eng: the original ctx (snapshots, statistics); cur: where the live points iterate
    try:
        if lam_pp is not None and lam_pp.ndim == 3:
            eng.set_lambda1_mask_k(lam[:, None, None] * lam_pp)        # single_admm_solver.py:114, per instance
        elif lam_pp is not None:
            # threshold (1/rho_k) * lambda1_k * mask: the kernel multiplies 1/rho_k by the (p,p) array, so
            # the per-instance lambda1 factor has to be the same for all k, or folded per instance
            assert np.all(lam == lam[0]), "a shared lambda1_mask needs one lambda1 (pass a (K,p,p) mask otherwise)"
            eng.set_lambda1_mask(lam[0] * lam_pp)
        if dims is not None:
            eng.set_instance_dims(dims)
        if hasattr(eng, "set_option"):
            eng.set_option("isolate", 1)
        pk = np.full(K, p) if dims is None else dims
        rhos = np.full(K, float(rho))
        done = np.zeros(K, dtype=bool)
        results = [None] * K
        last = np.zeros((K, 4))                                         # r_t, s_t, e_pri, e_dual of every point's last iteration
        dimk = (pk ** 2 + pk) / 2                                       # single_admm_solver.py:279, of the block itself
        keep_snapshots = selection_stats or latent
        cur, slots = eng, np.arange(K)                                   # slots[s]: the point in slot s of cur
        carried = np.zeros(K, dtype=np.int64)

        def state_of(s):
            k = slots[s]
            st = cur.state_k(s, latent)
            return st if dims is None else {nm: np.ascontiguousarray(A[:pk[k], :pk[k]]) for nm, A in st.items()}

        def finish(s, status, iters):
            k = slots[s]
            results[k] = (state_of(s), {'status': status, 'iterations': iters, 'rho': rhos[k]})
            if keep_snapshots:
                eng.snapshot_k(k) if cur is eng else eng.snapshot_from(k, cur, s)

        if hasattr(eng, "batch_run") and not verbose:
            # the whole loop in C (ggl_sgl_batch_run): decisions, X rescale, device snapshots of the points that finish,
            # parking of the ones that fail; Python is back in the picture only to compact the batch
            snaps, status, iters, why = _loop_in_c(eng, engines, K, 1, p, rhos, last, carried, dimk, tol, rtol, update_rho,
                                                   max_iter, compact, latent,
                                                   lambda sl: dict(lambda1=lam[sl], latent=latent,
                                                                   mu1=None if mu is None else mu[sl]), fetch=fetch)
            for k in range(K):
                sol = {nm: np.ascontiguousarray(A[k, :pk[k], :pk[k]]) for nm, A in snaps.items()}
                results[k] = (sol, {'status': status[k], 'iterations': int(iters[k]), 'rho': rhos[k]})
                if k in why:
                    results[k][1]['error'] = why[k]
        else:
            it = 0                                                           # batch iterations run so far
            while it < max_iter:
                sq = cur.sgl_batch_step(rhos[slots], lam[slots], latent, None if mu is None else mu[slots])
                carried[slots] += 1
                bad, newly, fac = _decide(sq, slots, rhos, done, last, dimk[slots], tol, rtol, update_rho, it, verbose,
                                          marked=_marked(cur, 1))
                if np.any(fac != 1.0):
                    cur.scale_X_batch(fac)      # (a failed point's factor is 1; single_admm_solver.py:205 comes before the break)
                it += 1
                # the converged points before the failed ones are parked: collecting them reads what the ctx knows about the
                # last L-step of the WHOLE batch, which parking a point must not disturb
                for s in newly:
                    finish(s, 'optimal', it)
                for s in bad:
                    # this point's data are not finite (a NaN in its S, a diverged iterate) or the library marked it (an
                    # eigensolver that did not converge): the reference's sequential walk (model_selection.py:619-633) would
                    # lose this point only -- so does the batch
                    finish(s, 'solver error', it)
                    results[slots[s]][1]['error'] = _why(cur, int(s), 1)
                    if hasattr(cur, "reset_instance"):
                        cur.reset_instance(int(s))
                if done.all():
                    break
                cur, slots = _compact(cur, slots, done, engines, compact, p=p, it=it - 1)
            for s, k in enumerate(slots):
                if results[k] is None:
                    finish(s, _leftover_status(last[k]), max_iter)
        for k in range(K):
            results[k][1]['carried'] = int(carried[k])
        ranks = None
        if latent:
            ranks, inconsistent = _final_L(eng, [results[k][0] for k in range(K)], 1)
            _mark_inconsistent(results, inconsistent)
        _late_failures(eng, results, 1)
        if selection_stats:
            assert dims is None, "selection statistics are taken over whole slots"
            st = eng.selection_stats()
            _late_failures(eng, results, 1)
            for k in range(K):
                results[k][1]['selection'] = {'Sdot': st[k, 0], 'logdet': st[k, 1], 'nnz': st[k, 2],
                                              'lambda_min': st[k, 3]}
            if latent:
                for k in range(K):
                    results[k][1]['selection']['rank'] = int(ranks[k])
            if tau_range is not None:
                tab, n_eig = eng.threshold_scan(tau_range)
                for k in range(K):
                    results[k][1]['selection']['threshold'] = tab[k].copy()
                    results[k][1]['selection']['threshold_eig_problems'] = n_eig
        # a point that ended as 'solver error' has no statistics: NaN in every table built from them (never the selected
        # point, never a rank of 0 that reads like a result) -- the reference's walk has no such point, it only warns
        for k in range(K):
            info = results[k][1]
            if info['status'] == 'solver error' and 'selection' in info:
                info['selection'] = {nm: (np.full_like(np.asarray(v, dtype=np.float64), np.nan) if nm == 'threshold' else
                                          (v if nm == 'threshold_eig_problems' else float('nan')))
                                     for nm, v in info['selection'].items()}
                info['selection']['failed'] = True
        if select is not None:
            k_sel, target = select(results)
            if k_sel is not None:
                own = results[int(k_sel)][0]
                if ('Omega' not in own or 'X' not in own) and fetch is not None and hasattr(eng, "snapshot_state_k"):
                    # (the C loop kept every finished point's state on the device and `fetch` left Omega / X there; the
                    # Python loop -- verbose=True, engines without batch_run -- downloaded them when the point finished and
                    # never took a state snapshot: ADVICE r5)
                    Om_k, X_k = eng.snapshot_state_k(int(k_sel))
                    q = int(pk[k_sel])
                    own.setdefault('Omega', np.ascontiguousarray(Om_k[:q, :q]))
                    own.setdefault('X', np.ascontiguousarray(X_k[:q, :q]))
                if target is not None:
                    for nm in ('Omega', 'X'):
                        if nm in own:
                            target.setdefault(nm, own[nm])
        _warn_failures(results)
    finally:
        for e in engines:
            e.close()
    return results


def _leftover_status(row):
    r_t, s_t, e_pri, e_dual = row
    return 'primal optimal' if r_t <= e_pri else ('dual optimal' if s_t <= e_dual else 'max iterations reached')


def _loop_in_c(eng, engines, n, group, p, rhos, last, carried, dims, tol, rtol, update_rho, max_iter, compact, latent, args_of,
               fetch=None):
    """The iteration loop of a batch with the host side in C (HipEngine.batch_run -> ggl_sgl_batch_run / ggl_mgl_batch_run):
    n points of ``group`` instances each; rhos (n,), last (n,4), carried (n,) are updated in place; ``args_of(slots)``: the
    keyword arguments of ``batch_run`` that describe the points in the slots of the live ctx.  Every point's solution is
    snapshotted on the device at the iteration it finishes (in the ORIGINAL ctx ``eng`` at its original index, also out of a
    compacted ctx) and all of them come back with one download per stack.  Returns ({'Omega','Theta','X'[,'L']} stacks of
    eng, status strings (n), iteration counts (n))."""
    status = np.zeros(n, dtype=np.int32)
    fin_iter = np.zeros(n, dtype=np.int32)
    reasons = {}
    cur, slots = eng, np.arange(n)
    inst = lambda sl: (sl[:, None] * group + np.arange(group)[None, :]).reshape(-1)     # instance slots of point slots
    dims = np.broadcast_to(np.asarray(dims, dtype=np.float64), (n,))
    it = 0
    while it < max_iter:
        ns = len(slots)
        # when to come back for a compaction: once enough slots of the live ctx hold finished points -- and never where the
        # cost model (_compact) cannot be met within max_iter
        stop_after = 0
        if compact and hasattr(cur, "subset") and \
                ns * group * COMPACT_SLOT_S_PER_P3 * float(p) ** 3 * max(max_iter, 10) >= COMPACT_COST_S:
            stop_after = max(COMPACT_MIN_DROP, int(np.ceil(COMPACT_FRACTION * ns)))
        rho_s, last_s = np.ascontiguousarray(rhos[slots]), np.ascontiguousarray(last[slots])
        st_s, fi_s = np.ascontiguousarray(status[slots]), np.ascontiguousarray(fin_iter[slots])
        done_before = int(np.count_nonzero(st_s))
        if stop_after:
            stop_after = max(stop_after, done_before + 1)
        k = cur.batch_run(max_iter - it, rho_s, last_s, st_s, fi_s, it, dims[slots], tol, rtol, update_rho,
                          snap=(eng, inst(slots)), stop_after=stop_after, **args_of(slots))
        for s in np.flatnonzero((st_s == 2) & (status[slots] != 2)):
            reasons[int(slots[s])] = _why(cur, int(s) * group, group)          # (asked of the ctx that marked it)
        rhos[slots], last[slots], status[slots], fin_iter[slots] = rho_s, last_s, st_s, fi_s
        carried[slots] += k
        it += k
        done = status != 0
        if done.all():
            break
        cur, slots = _compact(cur, slots, done, engines, compact, group=group, p=p, it=it - 1)
    for s, g in enumerate(slots):
        if status[g] == 0:                      # ran into max_iter: what it holds now is its solution
            for q in range(group):
                eng.snapshot_state_from(g * group + q, cur, s * group + q)
    names = {1: 'optimal', 2: 'solver error'}
    out_status = [names[int(status[g])] if status[g] else _leftover_status(last[g]) for g in range(n)]
    iters = np.where(status != 0, fin_iter, max_iter)
    return (eng.snapshots(latent) if fetch is None else eng.snapshots(latent, names=fetch)), out_status, iters, reasons


def _marked(eng, group):
    """(slots,) bool: problems with an instance the library has marked (GGL_OPT_ISOLATE: non-finite bounds, an eigensolver that
    did not converge, a failed eigendecomposition fallback of the L-step).  With isolation on such an instance no longer
    fails the call, and its numbers may well be finite -- the host has to ask."""
    if not hasattr(eng, "failed_instances"):
        return None
    f = np.asarray(eng.failed_instances()) != 0
    return f.reshape(-1, group).any(axis=1)


def _why(ctx, first, group):
    """Text for a 'solver error' point whose instances are first .. first + group - 1 of ctx: the library's reason for the
    first marked one, or that the point's own sums stopped being finite."""
    if hasattr(ctx, "failed_reason"):
        for q in range(group):
            r = ctx.failed_reason(first + q)
            if r:
                return r
    return "its residual sums are not finite (NaN / Inf in its data or a diverged iterate)"


def _warn_failures(results):
    """One warning per point that ends as 'solver error' (the reference warns about numerical trouble and goes on,
    solver/admm_solver.py:284-301): the point's index and what went wrong."""
    import warnings
    for g, res in enumerate(results):
        if res is not None and res[1].get('status') == 'solver error':
            warnings.warn(f"batch point {g}: solver error -- {res[1].get('error', 'marked by the library')}; "
                          f"the other points are not affected", RuntimeWarning, stacklevel=3)


def _mark_inconsistent(results, points):
    """Points whose rebuilt latent component is not the one their last iteration computed (_final_L): 'solver error'."""
    for g in points:
        if results[g] is not None and results[g][1]['status'] != 'solver error':
            results[g][1]['status'] = 'solver error'
            results[g][1]['error'] = ("the latent component rebuilt from the kept input of the last L-step differs from the "
                                      "iteration's own (an internal inconsistency: please report it)")


def _late_failures(eng, results, group):
    """Marks set on the ORIGINAL ctx after the iterations (the eigendecompositions of ggl_finalize_L and of the selection
    statistics run there over the snapshots of all points): such a point is reported as failed, not as 'optimal'."""
    m = _marked(eng, group)
    if m is None:
        return
    for g in np.flatnonzero(m):
        if results[g] is not None and results[g][1]['status'] != 'solver error':
            results[g][1]['status'] = 'solver error'
            results[g][1]['error'] = _why(eng, int(g) * group, group)


def _decide(sq, ids, rhos, done, last, dims, tol, rtol, update_rho, it, verbose, marked=None):
    """One iteration's host decisions for ALL points of a batch at once -- the reference's per-problem arithmetic
    (ADMM_stopping_criterion, solver/admm_solver.py:316-331; residual balancing, :227-233; what ``residuals_from_norms`` /
    ``next_rho`` do for one problem), element-wise in float64 in the same order of operations, so every decision is the one
    the per-point loop took.  A Python loop over the points costs ~12 us per point and iteration: for a 100-point grid of
    p = 64 problems that was 1.2 ms of host time per 0.1 ms of device time.

    sq: (len(ids), 5) squared norms of the slots; ids[s]: the point in slot s; marked: (len(ids),) bool, slots the library
    has marked as failed (they end like slots with non-finite sums).  Updates rhos / done / last (rows
    r_t, s_t, e_pri, e_dual) in place; returns (slots with non-finite sums, slots that converged now, X scaling factors)."""
    sq = np.asarray(sq, dtype=np.float64).reshape(len(ids), -1)
    live = ~done[ids]
    finite = np.all(np.isfinite(sq), axis=1)
    if marked is not None:
        finite &= ~np.asarray(marked, dtype=bool)         # (a marked point is treated like one with non-finite sums)
    bad = np.flatnonzero(live & ~finite)
    ok = live & finite
    n_om, n_thl, n_x, n_r, n_s = np.sqrt(np.where(ok[:, None], sq[:, :5], 0.0)).T
    rho = rhos[ids]
    r_t, s_t = n_r, rho * n_s
    e_pri = dims * tol + rtol * np.maximum(n_om, n_thl)
    e_dual = dims * tol + rtol * rho * n_x
    fac = np.ones(len(ids))
    if update_rho:
        rn = np.where(r_t >= 10 * s_t, 2 * rho, np.where(s_t >= 10 * r_t, 0.5 * rho, 1. * rho))
        fac[ok] = (rho / rn)[ok]
        rhos[ids[ok]] = rn[ok]
    last[ids[ok]] = np.stack([r_t, s_t, e_pri, e_dual], axis=1)[ok]
    if verbose:
        for s in np.flatnonzero(ok):
            print("%4d\t%3d\t%10.4g\t%10.4g\t%10.4g\t%10.4g" % (it, ids[s], r_t[s], s_t[s], e_pri[s], e_dual[s]))
    newly = np.flatnonzero(ok & (r_t <= e_pri) & (s_t <= e_dual))
    done[ids[bad]] = True
    done[ids[newly]] = True
    return bad, newly, fac


# Compaction of a batch of independent problems (the reference's walk, helper/model_selection.py:208-224,
# spends nothing on a point that has converged): once at least a quarter of the slots of the live ctx hold finished points,
# the points still iterating move to a smaller ctx (HipEngine.subset, device to device) and the products stop paying for the
# others.  The first step in the new ctx cannot speculate (no carried bounds) and the ctx itself costs a few allocations: the
# move is made only when at least COMPACT_MIN_DROP slots and COMPACT_FRACTION of the live ctx can be dropped.  Every info dict
# carries 'carried': the batch iterations the point occupied a slot for (its own 'iterations' + what it was dragged along).
COMPACT_FRACTION = 0.25
COMPACT_MIN_DROP = 2
# ... and only when the iterations it saves outweigh what the move costs.  A deterministic model, not a clock (the decision
# changes the last digits of the result -- the first step in the new ctx plans its schedule afresh -- and must not depend on
# the machine's mood): a slot of dimension p costs ~2.1e-13 * p^3 s of device time per iteration where the iteration is
# product-bound (p = 1000: 0.21 ms per slot, tools/bench_grid.py) and nothing below the launch-bound floor, the move ~8 ms
# (a ctx, its copies, its destruction, one step without speculation; at p = 500 a move that the model priced at 4 ms still
# lost: 33.5 vs 30.0 ms), and a batch that has run `it` iterations is taken to
# need as many again (at least 10).  Measured before the model (tools/bench_grid.py): compaction made a 20-point p = 50 grid
# 3 times and a 100-point p = 64 grid 4 times SLOWER (39.9 vs 13.4 ms, 60.2 vs 14.9 ms) while it takes 13 % off p = 1000.
COMPACT_SLOT_S_PER_P3 = 2.1e-13
COMPACT_COST_S = 8e-3


def _compact(cur, slots, done, engines, enabled, group=1, p=None, it=0):
    if not enabled or not hasattr(cur, "subset"):
        return cur, slots
    alive = np.flatnonzero(~done[slots])
    n_drop = len(slots) - len(alive)
    if len(alive) == 0 or n_drop < COMPACT_MIN_DROP or n_drop < COMPACT_FRACTION * len(slots):
        return cur, slots
    if p is not None and n_drop * group * COMPACT_SLOT_S_PER_P3 * float(p) ** 3 * max(it + 1, 10) < COMPACT_COST_S:
        return cur, slots
    idx = alive if group == 1 else (alive[:, None] * group + np.arange(group)[None, :]).reshape(-1)
    new = cur.subset(idx)
    engines.append(new)
    return new, slots[alive]


def _final_L(eng, sols, per_sol):
    """End of a latent batch: every snapshot's L is rebuilt from one eigendecomposition of its last L-step's input where that
    step was the sign iteration (``HipEngine.finalize_L``; solver/ggl_helper.py:29-36 is what the reference's callers get) and
    replaces the L downloaded at the iteration the problem converged.  sols[g]['L'] is (p,p) (per_sol = 1) or (per_sol,p,p).
    Returns the (len(sols) * per_sol,) ranks: the device's count of eigenvalues above the threshold where it rebuilt the
    instance, numpy's rule on the (eigendecomposition's) L otherwise."""
    n_rebuilt, rk = eng.finalize_L(1)
    rk = np.array(rk, dtype=np.int64)
    bad = []
    for g, sol in enumerate(sols):
        for k in range(per_sol):
            i = g * per_sol + k
            single = per_sol == 1 and sol['L'].ndim == 2
            L_it = sol['L'] if single else sol['L'][k]             # the L of the iteration the point finished in
            if rk[i] >= 0:
                L = eng.snapshot_L_k(i)
                q = sol['L'].shape[-1]                 # (instances of a padded batch are returned un-padded)
                L = np.ascontiguousarray(L[:q, :q])
                # the rebuilt L IS the iteration's L up to the sign iteration's residual: anything else means the kept input of that
                # L-step is not what the step saw (round 5's intermittent RANK table [[0,0,63],[0,0,108]] was of this kind) -- the
                # point is reported, never returned as if it were a result.  The residual is relative to the step's INPUT
                # C = Omega - Theta - X (~1e-13 |C|), not to L: an L that is exactly zero (every eigenvalue below the threshold)
                # comes out of the sign iteration as noise of that size (4e-12 seen at p = 16, tools/fuzz_parity.py ... stats), so
                # the floor scales with the iterate.  A lost input is off by the size of L itself, orders above either term.
                nrm = float(np.abs(L_it).max())
                scale = max([1.0] + [float(np.abs(sol[nm] if single else sol[nm][k]).max()) for nm in ('Omega', 'Theta', 'X')
                                     if nm in sol and np.all(np.isfinite(sol[nm] if single else sol[nm][k]))])
                if not (np.all(np.isfinite(L)) and float(np.abs(L - L_it).max()) <= 1e-6 * nrm + 1e-8 * scale):
                    bad.append(g)
                    continue
                if single:
                    sol['L'] = L
                else:
                    sol['L'][k] = L
            else:
                try:
                    rk[i] = _solver.latent_rank(L_it)
                except np.linalg.LinAlgError:          # (a NaN in L: the point is a failed one)
                    bad.append(g)
    for g in bad:
        rk[g * per_sol:(g + 1) * per_sol] = -1
    return rk, sorted(set(bad))


def pad_blocks(blocks, P, identity):
    """(p_k,p_k) arrays -> (K,P,P) stack: block k in the leading corner of slot k, an identity block (or zeros) behind it."""
    out = np.zeros((len(blocks), P, P))
    for k, B in enumerate(blocks):
        q = B.shape[0]
        out[k, :q, :q] = B
        if identity and q < P:
            d = np.arange(q, P)
            out[k, d, d] = 1.0
    return out


def ADMM_MGL_batch(S, lambda1, lambda2, reg, Omega_0=None, n_samples=None, tol=1e-5, rtol=1e-4, update_rho=True,
                   rho=1., max_iter=1000, verbose=False, latent=False, mu1=None, selection_stats=False,
                   tau_range=None, compact=True, fetch=None, select=None):
    """Solve ``ADMM_MGL(S, lambda1[g], lambda2[g], reg, Omega_0, ...)`` (solver/admm_solver.py:13-313) for every
    g of the 1-D arrays ``lambda1`` / ``lambda2`` at once: the G problems are the slabs of one (G*K,p,p) stack on
    the GPU, one batched Omega-step (and L-step) over all G*K matrices and one Theta-step launch per iteration.
    This is what the MAIN LOOP of the reference's ``grid_search`` (helper/model_selection.py:208-224) does one
    (lambda1, lambda2) point after the other.

    Every problem keeps its OWN rho, residuals, rho updates and stopping decision, exactly as if it had been solved
    on its own from ``Omega_0`` (default: identity; Theta_0 = Omega_0, X_0 = 0 as in admm_solver.py:142-150), so each
    returned (sol, info) equals the independent solve; a problem's solution is snapshotted at the iteration it
    converges; once a quarter of the live slots hold finished problems the rest go on in a smaller stack (``compact``).  A
    point whose data are not finite (a NaN in its S, a diverged iterate) ends with ``info['status'] == 'solver error'`` and
    costs that point only, as in the reference's sequential walk (the ctx runs with GGL_OPT_ISOLATE).
    mu1: (K,) shared by all problems or (G,K); n_samples as in ADMM_MGL.
    Returns a list of G ``(sol, info)``; ``info`` carries 'status', 'iterations', 'rho' (+ 'selection': per-instance
    (K,4) array of <S,Theta>, log det Theta, count_nonzero(Theta), lambda_min(Theta) from the GPU when
    ``selection_stats``; with ``latent`` 'rank': (K,) numpy.linalg.matrix_rank of the problem's L_k, see ADMM_SGL_batch)."""
    S = as_c(S)
    assert S.ndim == 3 and S.shape[1] == S.shape[2]
    assert reg in ['GGL', 'FGL']
    K, p, _ = S.shape
    lam1 = as_c(np.atleast_1d(lambda1)).reshape(-1)
    lam2 = as_c(np.atleast_1d(lambda2)).reshape(-1)
    assert len(lam1) == len(lam2)
    G = len(lam1)
    assert min(lam1.min(), lam2.min()) > 0
    assert rho > 0, "ADMM penalization parameter must be positive."
    if latent:
        assert mu1 is not None
        mu = np.asarray(mu1, dtype=np.float64)
        if mu.ndim == 0:
            mu = mu * np.ones(K)
        mu = as_c(np.broadcast_to(mu, (G, K))).reshape(-1)
        assert np.all(mu > 0)
    else:
        mu = None
    if n_samples is None:
        nk = None
    elif isinstance(n_samples, (int, np.integer)):
        nk = float(n_samples) * np.ones(K)
    else:
        nk = as_c(n_samples).reshape(-1)
        assert len(nk) == K
    Om0 = np.repeat(np.eye(p)[None], K, axis=0) if Omega_0 is None else as_c(Omega_0)
    assert Om0.shape == S.shape
    # (G,K,p,p) broadcast VIEWS of the one problem's stacks, and of one zero matrix: uploaded once, replicated on the device
    rep = lambda A: np.broadcast_to(as_c(A), (G,) + A.shape)
    eng = _solver.ENGINE(rep(S), rep(Om0), rep(Om0), np.broadcast_to(np.zeros((p, p)), (G * K, p, p)))
    engines = [eng]
    try:
        if hasattr(eng, "set_option"):
            eng.set_option("isolate", 1)
        rhos = np.full(G, float(rho))
        done = np.zeros(G, dtype=bool)
        results = [None] * G
        last = np.zeros((G, 4))
        dim = K * ((p ** 2 + p) / 2)
        cur, slots = eng, np.arange(G)                                   # slots[s]: the problem in slot s of cur
        carried = np.zeros(G, dtype=np.int64)
        inst = lambda v: None if v is None else v.reshape(G, K)[slots].reshape(-1)

        def collect(s, status, iters):
            g = slots[s]
            parts = [cur.state_k(s * K + k, True) for k in range(K)]
            sol = {nm: np.stack([q[nm] for q in parts]) for nm in ('Omega', 'Theta', 'L', 'X')}
            results[g] = (sol, {'status': status, 'iterations': iters, 'rho': rhos[g]})
            if selection_stats or latent:
                for k in range(K):
                    eng.snapshot_k(g * K + k) if cur is eng else eng.snapshot_from(g * K + k, cur, s * K + k)

        if hasattr(eng, "batch_run") and not verbose:
            snaps, status, iters, why = _loop_in_c(eng, engines, G, K, p, rhos, last, carried, float(dim), tol, rtol, update_rho,
                                              max_iter, compact, True,
                                              lambda sl: dict(lambda1=lam1[sl], lambda2=lam2[sl], reg=reg, latent=latent,
                                                              mu1=None if mu is None else mu.reshape(G, K)[sl].reshape(-1),
                                                              nk=nk, G=len(sl)), fetch=fetch)
            for g in range(G):
                sol = {nm: snaps[nm][g * K:(g + 1) * K].copy() for nm in ('Omega', 'Theta', 'L', 'X') if nm in snaps}
                results[g] = (sol, {'status': status[g], 'iterations': int(iters[g]), 'rho': rhos[g]})
                if g in why:
                    results[g][1]['error'] = why[g]
        else:
            it = 0
            while it < max_iter:
                sq = cur.mgl_batch_step(len(slots), rhos[slots], lam1[slots], lam2[slots], reg, latent, inst(mu), nk)
                carried[slots] += 1
                bad, newly, fac = _decide(sq, slots, rhos, done, last, dim, tol, rtol, update_rho, it, verbose,
                                          marked=_marked(cur, K))
                if np.any(fac != 1.0):
                    cur.scale_X_batch(np.repeat(fac, K))
                it += 1
                for s in newly:                      # (before the failed ones are parked: see ADMM_SGL_batch)
                    collect(s, 'optimal', it)
                for s in bad:
                    # (see ADMM_SGL_batch: a point with non-finite data or a mark costs that point only)
                    collect(s, 'solver error', it)
                    results[slots[s]][1]['error'] = _why(cur, int(s) * K, K)
                    if hasattr(cur, "reset_instance"):
                        for k in range(K):
                            cur.reset_instance(int(s) * K + k)
                if done.all():
                    break
                cur, slots = _compact(cur, slots, done, engines, compact, group=K, p=p, it=it - 1)
            for s, g in enumerate(slots):
                if results[g] is None:
                    collect(s, _leftover_status(last[g]), max_iter)
        for g in range(G):
            results[g][1]['carried'] = int(carried[g])
        if latent:
            rk, inconsistent = _final_L(eng, [results[g][0] for g in range(G)], K)
            _mark_inconsistent(results, inconsistent)
            for g in range(G):
                results[g][1]['rank'] = rk[g * K:(g + 1) * K].astype(np.float64)
        if selection_stats:
            st = eng.selection_stats()
            for g in range(G):
                results[g][1]['selection'] = st[g * K:(g + 1) * K].copy()
            if tau_range is not None:
                tab, _ = eng.threshold_scan(tau_range)
                for g in range(G):
                    results[g][1]['threshold'] = tab[g * K:(g + 1) * K].copy()
        _late_failures(eng, results, K)
        for g in range(G):                      # (see ADMM_SGL_batch: a failed point has no statistics)
            info = results[g][1]
            if info['status'] == 'solver error':
                for nm in ('selection', 'threshold', 'rank'):
                    if nm in info:
                        info[nm] = np.full(np.shape(info[nm]), np.nan)
        if select is not None:
            # (fetch / select as in ADMM_SGL_batch: the whole solution of the ONE point the caller selects)
            g_sel, target = select(results)
            if g_sel is not None:
                own = results[int(g_sel)][0]
                if ('Omega' not in own or 'X' not in own) and fetch is not None and hasattr(eng, "snapshot_state_k"):
                    parts = [eng.snapshot_state_k(int(g_sel) * K + k) for k in range(K)]
                    own.setdefault('Omega', np.stack([q[0] for q in parts]))
                    own.setdefault('X', np.stack([q[1] for q in parts]))
                if target is not None:
                    for nm in ('Omega', 'X'):
                        if nm in own:
                            target.setdefault(nm, own[nm])
        _warn_failures(results)
    finally:
        for e in engines:
            e.close()
    return results
