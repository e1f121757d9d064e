"""Batches of independent problems (the grids of model selection): a point that fails costs that point, a point that has
converged costs nothing more.  Reference: the grid walks of helper/model_selection.py:208-224 and :619-633
solve their points one after the other -- a NaN in one point's data never reaches another point, and a finished point is
finished.  Here the points share launches, so both properties have to be built: GGL_OPT_ISOLATE / ggl_reset_instance and
ggl_ctx_create_subset (include/ggl_hip.h)."""
import contextlib
import io

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("p,latent", [(40, False), (160, False), (160, True)])
def test_sgl_batch_with_one_poisoned_point(p, latent):
    """K = 6 single problems in one stack, S of point 2 holds a NaN: that point reports 'solver error', the other five equal
    their independent solves (LDS-Jacobi route at p = 40, Newton-Schulz route at p = 160, with and without latent variables)."""
    from gglasso_amd import synth, ADMM_SGL
    from gglasso_amd.batch import ADMM_SGL_batch
    K = 6
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=77 + p)
    bad = S.copy()
    bad[2, 3, 5] = bad[2, 5, 3] = np.nan
    lam = np.linspace(0.05, 0.2, K)
    kw = dict(tol=1e-8, rtol=1e-8, max_iter=300)
    if latent:
        kw.update(latent=True, mu1=0.6)
    with pytest.warns(RuntimeWarning, match="batch point 2: solver error") as rec:
        res = ADMM_SGL_batch(bad, lam, **kw)
    assert res[2][1]['status'] == 'solver error'
    # the point comes with the reason (ggl_failed_reason, or its own non-finite sums), and it is the only one warned about
    assert isinstance(res[2][1].get('error'), str) and len(res[2][1]['error']) > 10, res[2][1]
    assert len([w for w in rec if "solver error" in str(w.message)]) == 1
    assert all('error' not in res[k][1] for k in range(K) if k != 2)
    for k in range(K):
        if k == 2:
            continue
        sol, info = quiet(ADMM_SGL, S[k], lam[k], np.eye(p), **kw)
        assert res[k][1]['status'] == info['status'] == 'optimal', (k, res[k][1])
        assert np.all(np.isfinite(res[k][0]['Theta']))
        assert np.abs(res[k][0]['Theta'] - sol['Theta']).max() <= 1e-9
        if latent:
            assert np.abs(res[k][0]['L'] - sol['L']).max() <= 1e-9
            assert np.linalg.matrix_rank(res[k][0]['L']) == np.linalg.matrix_rank(sol['L'])


@pytest.mark.parametrize("reg,p", [("GGL", 150), ("FGL", 40)])
def test_mgl_batch_step_isolates_a_poisoned_problem(reg, p):
    """Engine level, G = 3 problems of K = 4 instances in one stack, one entry of problem 1's S is NaN.  With GGL_OPT_ISOLATE the
    step returns (non-finite sums for problem 1, and) for problems 0 and 2 exactly the sums of the healthy batch; after
    ggl_reset_instance on problem 1's slots every later step is finite everywhere and the healthy problems' iterates are
    untouched.  Without the option the same step is an error for the whole batch (p > 128: the planner sees the NaN bound)."""
    from gglasso_amd import solver, synth
    G, K = 3, 4
    S, _ = synth.make_problem(reg, K, p, N=2 * p, seed=5)
    stack = np.concatenate([S] * G)
    bad = stack.copy()
    bad[K + 1, 0, 1] = bad[K + 1, 1, 0] = np.nan
    eye = np.repeat(np.eye(p)[None], G * K, axis=0)
    rho, l1, l2 = np.ones(G), np.full(G, 0.05), np.array([0.01, 0.02, 0.03])

    def run(Sx, isolate, steps):
        eng = solver.HipEngine(Sx, eye, eye, np.zeros_like(eye), options={"isolate": isolate})
        try:
            sums = []
            for it in range(steps):
                sq = eng.mgl_batch_step(G, rho, l1, l2, reg, False, None, None)
                sums.append(sq.copy())
                if it == 0 and isolate and not np.all(np.isfinite(sq[1])):
                    marked = eng.failed_instances()
                    for k in range(K):
                        eng.reset_instance(K + k)
            return np.array(sums), eng.state(), (marked if isolate else None)
        finally:
            eng.close()

    good_sums, good_state, _ = run(stack, 0, 4)
    sums, state, marked = run(bad, 1, 4)
    assert not np.all(np.isfinite(sums[0, 1]))
    if p > 128:
        assert marked[K + 1] == 1 and marked[:K].sum() == 0 and marked[2 * K:].sum() == 0, marked
    for g in (0, 2):
        assert np.allclose(sums[:, g], good_sums[:, g], rtol=1e-9, atol=0)
        sl = slice(g * K, (g + 1) * K)
        for nm in ("Omega", "Theta", "X"):
            assert np.abs(state[nm][sl] - good_state[nm][sl]).max() <= 1e-10, (g, nm)
    assert np.all(np.isfinite(sums[1:]))                       # the parked problem iterates on the identity problem
    if p > 128:
        with pytest.raises(RuntimeError):
            run(bad, 0, 1)


def test_mgl_grid_with_a_poisoned_point_through_the_batch_driver(monkeypatch):
    """ADMM_MGL_batch: the grid's points share S, so the fault is injected into ONE point's slab of the stack after the upload
    (a NaN written into its S on the device); the driver must hand back G - 1 good solutions and one 'solver error'."""
    from gglasso_amd import solver, synth, ADMM_MGL
    from gglasso_amd.batch import ADMM_MGL_batch
    K, p, G = 3, 140, 4
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=9)
    lam1, lam2 = np.array([0.2, 0.1, 0.07, 0.05]), np.array([0.05, 0.03, 0.02, 0.01])
    real = solver.HipEngine

    class Poisoned(real):
        def __init__(self, Sx, *a, **kw):
            Sx = np.array(np.broadcast_to(Sx, (G, K, p, p)))          # materialise the shared stack, poison point 2
            Sx[2, 1, 4, 7] = Sx[2, 1, 7, 4] = np.nan
            super().__init__(Sx, *a, **kw)

    monkeypatch.setattr(solver, "ENGINE", Poisoned)
    res = ADMM_MGL_batch(S, lam1, lam2, "GGL", tol=1e-8, rtol=1e-8, max_iter=400)
    monkeypatch.setattr(solver, "ENGINE", real)
    assert [r[1]['status'] for r in res] == ['optimal', 'optimal', 'solver error', 'optimal']
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    for g in (0, 1, 3):
        sol, info = quiet(ADMM_MGL, S, lam1[g], lam2[g], "GGL", Om0, tol=1e-8, rtol=1e-8, max_iter=400)
        assert np.abs(res[g][0]['Theta'] - sol['Theta']).max() <= 1e-9, g


@pytest.mark.parametrize("p,latent", [(60, False), (200, False), (200, True)])
def test_sgl_batch_compaction_equals_the_uncompacted_batch(p, latent, monkeypatch):
    """A lambda1 path whose points need very different iteration counts, with and without compaction: same status and iteration
    count per point, same solutions (1e-9: a step in a fresh ctx is planned from its own bounds, not from carried ones), and the
    compacted run carries its points for far fewer batch iterations.  (The cost model that keeps small problems from being
    compacted at all -- batch.COMPACT_COST_S -- is switched off: this is the mechanism's test.)"""
    from gglasso_amd import batch, synth
    from gglasso_amd.batch import ADMM_SGL_batch
    monkeypatch.setattr(batch, "COMPACT_COST_S", 0.0)
    S, _ = synth.make_problem("GGL", 1, p, N=2 * p, seed=21)
    lam = np.logspace(0, -2, 12)
    kw = dict(tol=1e-7, rtol=1e-7, max_iter=500, selection_stats=True)
    if latent:
        kw.update(latent=True, mu1=np.linspace(0.4, 1.5, 12))
    a = ADMM_SGL_batch(S[0], lam, compact=False, **kw)
    b = ADMM_SGL_batch(S[0], lam, compact=True, **kw)
    its = np.array([r[1]['iterations'] for r in a])
    assert its.max() > its.min() + 4 and np.array_equal(its, [r[1]['iterations'] for r in b]), its
    for k in range(len(lam)):
        assert a[k][1]['status'] == b[k][1]['status'] == 'optimal'
        for nm in a[k][0]:
            assert np.abs(a[k][0][nm] - b[k][0][nm]).max() <= 1e-9, (k, nm)
        for nm in ('Sdot', 'logdet', 'nnz'):
            assert np.isclose(a[k][1]['selection'][nm], b[k][1]['selection'][nm], rtol=1e-9), (k, nm)
        if latent:
            assert a[k][1]['selection']['rank'] == b[k][1]['selection']['rank'] == np.linalg.matrix_rank(b[k][0]['L'])
    carried_a = sum(r[1]['carried'] for r in a)
    carried_b = sum(r[1]['carried'] for r in b)
    assert carried_a == len(lam) * its.max()
    assert its.sum() <= carried_b < carried_a, (carried_a, carried_b, its.sum())


def test_mgl_batch_compaction_equals_the_uncompacted_batch(monkeypatch):
    from gglasso_amd import batch, synth
    from gglasso_amd.batch import ADMM_MGL_batch
    monkeypatch.setattr(batch, "COMPACT_COST_S", 0.0)
    K, p = 3, 150
    S, _ = synth.make_problem("FGL", K, p, N=2 * p, seed=23)
    lam1 = np.logspace(-0.3, -1.7, 8)
    lam2 = np.full(8, 0.02)
    kw = dict(tol=1e-7, rtol=1e-7, max_iter=500, latent=True, mu1=np.array([0.8, 0.6, 0.7]), selection_stats=True)
    a = ADMM_MGL_batch(S, lam1, lam2, "FGL", compact=False, **kw)
    b = ADMM_MGL_batch(S, lam1, lam2, "FGL", compact=True, **kw)
    its = np.array([r[1]['iterations'] for r in a])
    assert np.array_equal(its, [r[1]['iterations'] for r in b]), (its, [r[1]['iterations'] for r in b])
    for g in range(8):
        for nm in ('Omega', 'Theta', 'L', 'X'):
            assert np.abs(a[g][0][nm] - b[g][0][nm]).max() <= 1e-9, (g, nm)
        assert np.array_equal(a[g][1]['rank'], b[g][1]['rank'])
        assert np.allclose(a[g][1]['selection'], b[g][1]['selection'], rtol=1e-9)
    assert sum(r[1]['carried'] for r in b) < sum(r[1]['carried'] for r in a)


def test_small_problems_are_not_compacted():
    """The move to a smaller ctx costs ~8 ms; an iteration of a 12-point p = 60 batch costs 0.1 ms.  By default (the cost model
    of batch._compact) such a batch stays where it is: every point is carried to the end, and the results are the
    uncompacted ones bit for bit."""
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_SGL_batch
    S, _ = synth.make_problem("GGL", 1, 60, N=120, seed=21)
    lam = np.logspace(0, -2, 12)
    kw = dict(tol=1e-7, rtol=1e-7, max_iter=500)
    a = ADMM_SGL_batch(S[0], lam, compact=False, **kw)
    b = ADMM_SGL_batch(S[0], lam, compact=True, **kw)
    assert [r[1]['carried'] for r in a] == [r[1]['carried'] for r in b]
    for k in range(len(lam)):
        assert np.array_equal(a[k][0]['Theta'], b[k][0]['Theta'])


def test_ext_grid_with_a_poisoned_point(monkeypatch):
    """ext_ADMM_MGL_batch (the non-conforming grid of grid_search, helper/model_selection.py:208-224 with solver = ext_ADMM_MGL):
    a NaN written into ONE grid point's slab of the padded stack -- that point reports 'solver error', the other five end
    where their own ext_ADMM_MGL solves end."""
    import ext_checks
    from gglasso_amd import solver, ext_solver
    g = load_golden("g14_ext_admm_nonconforming")
    K, p, S, G, Om0 = ext_checks.g14_inputs(g)
    l1s, l2s = [0.2, 0.08, 0.03], [0.1, 0.02]
    pts = [(a, b) for b in l2s for a in l1s]
    real = solver.HipEngine

    class Poisoned(real):
        def __init__(self, Sx, *a, **kw):
            Sx = np.array(Sx)
            Sx[4 * K + 1, 0, 1] = Sx[4 * K + 1, 1, 0] = np.nan          # grid point 4, instance 1
            super().__init__(Sx, *a, **kw)

    monkeypatch.setattr(solver, "ENGINE", Poisoned)
    res = quiet(ext_solver.ext_ADMM_MGL_batch, S, [a for a, _ in pts], [b for _, b in pts], 'GGL', G, tol=1e-8, rtol=1e-8)
    monkeypatch.setattr(solver, "ENGINE", real)
    assert [r[1]['status'] for r in res] == ['optimal'] * 4 + ['solver error', 'optimal']
    for i, (a, b) in enumerate(pts):
        if i == 4:
            continue
        ref, rinfo = quiet(ext_solver.ext_ADMM_MGL, S, a, b, 'GGL', {k: v.copy() for k, v in Om0.items()}, G, tol=1e-8, rtol=1e-8)
        for k in range(K):
            assert np.abs(res[i][0]['Theta'][k] - ref['Theta'][k]).max() <= 1e-9, (i, k)


def test_kept_L_step_input_survives_parking_and_compaction():
    """whether the last L-step was the sign iteration (and its kept input C) is state of the whole ctx.  Parking a
    failed instance (ggl_reset_instance) must not make the OTHER instances' snapshots lose it, and a compacted ctx
    (ggl_ctx_create_subset) must carry it, so that a point collected before the new ctx's first L-step is still rebuilt by
    ggl_finalize_L (rank from the eigendecomposition, >= 0) instead of keeping the iteration's L."""
    from gglasso_amd import solver, synth
    K, p = 4, 150
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=3)
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    eng = solver.HipEngine(S, eye, eye, np.zeros_like(eye), options={"isolate": 1})
    sub = None
    try:
        rho, lam, mu = np.ones(K), np.full(K, 0.1), np.full(K, 0.5)
        for _ in range(6):
            eng.sgl_batch_step(rho, lam, True, mu)
        eng.reset_instance(1)
        eng.snapshot_k(0)
        sub = eng.subset(np.array([2, 3]))
        eng.snapshot_from(3, sub, 1)                      # before any step in the new ctx
        n, rk = eng.finalize_L(1)
        assert n == 2 and rk[0] >= 0 and rk[3] >= 0 and rk[1] == -1 and rk[2] == -1
        for k in (0, 3):
            L = eng.snapshot_L_k(k)
            assert np.linalg.matrix_rank(L) == rk[k]
    finally:
        if sub is not None:
            sub.close()
        eng.close()


def test_marked_instance_with_finite_sums_is_reported(monkeypatch):
    """with GGL_OPT_ISOLATE a non-converged eigensolver marks the instance and the call goes on; the sums of that
    point can be finite.  The batch drivers ask the library (ggl_failed_instances) after every step: a marked point ends as
    'solver error', never as 'optimal'."""
    from gglasso_amd import solver, synth
    from gglasso_amd.batch import ADMM_SGL_batch
    K, p = 4, 30
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=9)
    real = solver.HipEngine.failed_instances
    calls = {"n": 0}

    def fake(self):
        calls["n"] += 1
        out = real(self)
        if calls["n"] >= 3 and self.K == K:
            out[1] = 1                                   # as if instance 1's eigensolver had not converged in step 3
        return out

    monkeypatch.setattr(solver.HipEngine, "failed_instances", fake)
    # (verbose: the driver's Python loop, which asks through HipEngine.failed_instances -- the C loop of ggl_sgl_batch_run reads
    # the ctx's marks directly and hands them to the same decision function, tests/test_cpu_batch_decisions.py)
    res = quiet(ADMM_SGL_batch, S, np.full(K, 0.1), tol=1e-8, rtol=1e-8, max_iter=200, compact=False, verbose=True)
    assert res[1][1]['status'] == 'solver error' and res[1][1]['iterations'] == 3
    assert all(res[k][1]['status'] == 'optimal' for k in (0, 2, 3))


@pytest.mark.parametrize("p,latent", [(40, False), (150, True)])
def test_sgl_batch_c_loop_is_the_python_loop(p, latent):
    """The batch's host loop in C (ggl_sgl_batch_run: per-point stopping test, rho rule, X rescale, device snapshots, one
    download per stack) against the drivers' Python loop (verbose=True; the reference's arithmetic in NumPy, one download per
    point): the same kernels in the same order under the same decisions -- statuses, iteration counts, rhos and every
    solution array bit for bit (single_admm_solver.py:186-214; the walk: helper/model_selection.py:619-633)."""
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_SGL_batch
    K = 7
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=11 + p)
    lam = np.logspace(-0.5, -1.5, K)
    kw = dict(tol=1e-7, rtol=1e-6, max_iter=60, compact=False)
    if latent:
        kw.update(latent=True, mu1=0.7)
    a = ADMM_SGL_batch(S, lam, **kw)
    b = quiet(ADMM_SGL_batch, S, lam, verbose=True, **kw)
    assert {i['status'] for _, i in a} <= {'optimal', 'max iterations reached', 'primal optimal', 'dual optimal'}
    for (sa, ia), (sb, ib) in zip(a, b):
        assert (ia['status'], ia['iterations'], ia['rho'], ia['carried']) == (ib['status'], ib['iterations'], ib['rho'], ib['carried'])
        assert sa.keys() == sb.keys()
        for nm in sa:
            assert np.array_equal(sa[nm], sb[nm]), nm


def test_mgl_batch_c_loop_is_the_python_loop():
    """The same for G multiple-graph problems in one stack (ggl_mgl_batch_run; admm_solver.py:215-237, the grid walk
    helper/model_selection.py:208-224), with compaction on both sides."""
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_MGL_batch
    K, p = 3, 60
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=5)
    l1 = np.array([0.2, 0.1, 0.05, 0.2, 0.1, 0.05])
    l2 = np.array([0.05, 0.05, 0.05, 0.01, 0.01, 0.01])
    kw = dict(tol=1e-7, rtol=1e-6, max_iter=80)
    a = ADMM_MGL_batch(S, l1, l2, "GGL", **kw)
    b = quiet(ADMM_MGL_batch, S, l1, l2, "GGL", verbose=True, **kw)
    for (sa, ia), (sb, ib) in zip(a, b):
        assert (ia['status'], ia['iterations'], ia['rho']) == (ib['status'], ib['iterations'], ib['rho'])
        for nm in ('Omega', 'Theta', 'L', 'X'):
            assert np.array_equal(sa[nm], sb[nm]), nm


def test_batch_run_without_snapshots_returns_at_the_first_event():
    """ggl_sgl_batch_run with snap_ctx = NULL: the call comes back after the first iteration in which a live point converges
    (status 1, fin_iter = it_base + that iteration), rho / last updated in place; the caller collects and calls again."""
    from gglasso_amd import solver, synth
    K, p = 5, 30
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=2)
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    eng = solver.HipEngine(S, eye, eye, np.zeros_like(eye))
    try:
        rho, last = np.ones(K), np.zeros((K, 4))
        status, fin = np.zeros(K, dtype=np.int32), np.zeros(K, dtype=np.int32)
        lam = np.array([0.5, 0.3, 0.2, 0.1, 0.05])
        dims = np.full(K, (p * p + p) / 2)
        n = eng.batch_run(500, rho, last, status, fin, 100, dims, 1e-7, 1e-5, True, lambda1=lam)
        assert 1 <= n < 500 and np.count_nonzero(status == 1) >= 1 and np.all(status != 2)
        done = status == 1
        assert np.all(fin[done] == 100 + n) and np.all(fin[~done] == 0)
        assert np.all(last[done, 0] <= last[done, 2]) and np.all(last[done, 1] <= last[done, 3])
        n2 = eng.batch_run(500, rho, last, status, fin, 100 + n, dims, 1e-7, 1e-5, True, lambda1=lam)
        assert n2 >= 1 and np.all(fin[done] == 100 + n)          # finished points are dragged along, untouched
    finally:
        eng.close()
