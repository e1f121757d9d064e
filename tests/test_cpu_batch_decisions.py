"""Host logic of the batch drivers (no GPU): ``batch._decide`` takes one iteration's stopping / rho decisions for all points
of a batch at once and must take, bit for bit, the decisions of the reference's per-problem arithmetic
(ADMM_stopping_criterion, solver/admm_solver.py:316-331; residual balancing :227-233 -- ``solver.residuals_from_norms`` /
``solver.next_rho``)."""
import numpy as np

from gglasso_amd import batch, solver


def _loop(sq, ids, rhos, done, last, dims, tol, rtol, update_rho):
    bad, newly, fac = [], [], np.ones(len(ids))
    for s, k in enumerate(ids):
        if done[k]:
            continue
        if not np.all(np.isfinite(sq[s])):
            bad.append(s)
            done[k] = True
            continue
        r_t, s_t, e_pri, e_dual = solver.residuals_from_norms(sq[s], rhos[k], tol, rtol, dims[s])
        if update_rho:
            rn = solver.next_rho(rhos[k], r_t, s_t)
            fac[s] = rhos[k] / rn
            rhos[k] = rn
        last[k] = (r_t, s_t, e_pri, e_dual)
        if (r_t <= e_pri) and (s_t <= e_dual):
            done[k] = True
            newly.append(s)
    return np.array(bad, dtype=np.int64), np.array(newly, dtype=np.int64), fac


def test_vectorised_decisions_are_the_per_point_ones():
    rng = np.random.default_rng(0)
    for trial in range(200):
        n_all = int(rng.integers(1, 40))
        ids = np.sort(rng.choice(n_all, size=int(rng.integers(1, n_all + 1)), replace=False))
        sq = rng.uniform(0, 1, (len(ids), 5)) ** rng.integers(1, 12)            # residuals from O(1) down to ~1e-12
        sq[:, :3] *= rng.uniform(1, 1e4, (len(ids), 1))
        # the boundaries of the rho rule and of the stopping test, exactly
        if len(ids) > 2:
            sq[0, 3] = 100.0 * sq[0, 4]
            sq[1, 4] = 100.0 * sq[1, 3]
        if trial % 5 == 0:
            sq[rng.integers(len(ids)), rng.integers(5)] = [np.nan, np.inf, -np.inf][trial % 3]
        rhos0 = 2.0 ** rng.integers(-4, 5, n_all).astype(float)
        done0 = rng.uniform(size=n_all) < 0.2
        dims = (rng.integers(2, 60, len(ids)) ** 2).astype(float)
        tol, rtol = 10.0 ** -rng.integers(3, 9), 10.0 ** -rng.integers(2, 8)
        for update_rho in (True, False):
            a = dict(rhos=rhos0.copy(), done=done0.copy(), last=np.zeros((n_all, 4)))
            b = dict(rhos=rhos0.copy(), done=done0.copy(), last=np.zeros((n_all, 4)))
            ra = batch._decide(sq, ids, a["rhos"], a["done"], a["last"], dims, tol, rtol, update_rho, 0, False)
            rb = _loop(sq, ids, b["rhos"], b["done"], b["last"], dims, tol, rtol, update_rho)
            for x, y in zip(ra, rb):
                assert np.array_equal(np.asarray(x), np.asarray(y))
            for key in ("rhos", "done", "last"):
                assert np.array_equal(a[key], b[key]), key


def test_scalar_dimension_as_in_the_multiple_graph_batch():
    sq = np.array([[4.0, 9.0, 1.0, 1e-12, 1e-14], [4.0, 9.0, 1.0, 1.0, 1e-4]])
    ids = np.arange(2)
    rhos, done, last = np.ones(2), np.zeros(2, dtype=bool), np.zeros((2, 4))
    bad, newly, fac = batch._decide(sq, ids, rhos, done, last, 50.0, 1e-7, 1e-5, True, 0, False)
    assert len(bad) == 0 and list(newly) == [0] and list(done) == [True, False]
    assert list(rhos) == [2.0, 2.0] and list(fac) == [0.5, 0.5]      # r_t >= 10 s_t for both


def test_compaction_cost_model():
    """batch._compact moves the live points to a smaller ctx only where the saved iterations outweigh the move (a model in p,
    the dropped slots and the iterations so far -- never a clock: the decision shows in the last digits of the result)."""
    class Eng:
        def subset(self, idx):
            return ("subset", tuple(int(i) for i in idx))

    def moved(p, n, n_done, it, group=1):
        done = np.zeros(n, dtype=bool)
        done[:n_done] = True
        cur, slots = batch._compact(Eng(), np.arange(n), done, [], True, group=group, p=p, it=it)
        return not isinstance(cur, Eng)

    assert moved(1000, 20, 5, 3)                    # C2 of SURVEY section 8: compacts after a few iterations
    assert not moved(64, 100, 60, 50)               # a hundred small problems: never
    assert not moved(50, 20, 10, 80)
    assert not moved(500, 20, 5, 3) and not moved(500, 20, 10, 18) and moved(500, 20, 10, 40)
    assert moved(500, 12, 6, 10, group=6) and not moved(500, 12, 6, 10, group=1)
    assert not moved(1000, 20, 1, 3) and not moved(1000, 20, 4, 3)      # the old rules stay: >= 2 slots and a quarter of the ctx


def test_c_decisions_are_the_numpy_ones():
    """ggl_batch_decide -- what ggl_sgl_batch_run / ggl_mgl_batch_run take between two iterations without returning to Python --
    against ``batch._decide`` bit for bit: rhos, factors, the stored residuals, which points converged and which failed,
    including the rule's boundaries, non-finite sums and points the library marked (host only: no GPU needed)."""
    import ctypes
    from gglasso_amd import _lib
    lib = _lib.load()
    ptr = _lib.ptr
    rng = np.random.default_rng(1)
    for trial in range(300):
        n = int(rng.integers(1, 40))
        ids = np.arange(n)
        sq = rng.uniform(0, 1, (n, 5)) ** rng.integers(1, 12)
        sq[:, :3] *= rng.uniform(1, 1e4, (n, 1))
        if n > 2:
            sq[0, 3] = 100.0 * sq[0, 4]
            sq[1, 4] = 100.0 * sq[1, 3]
        if trial % 5 == 0:
            sq[rng.integers(n), rng.integers(5)] = [np.nan, np.inf, -np.inf][trial % 3]
        rhos0 = 2.0 ** rng.integers(-4, 5, n).astype(float)
        done0 = rng.uniform(size=n) < 0.2
        marked = rng.uniform(size=n) < (0.1 if trial % 3 == 0 else 0.0)
        dims = (rng.integers(2, 60, n) ** 2).astype(float)
        tol, rtol = 10.0 ** -rng.integers(3, 9), 10.0 ** -rng.integers(2, 8)
        for update_rho in (True, False):
            a = dict(rhos=rhos0.copy(), done=done0.copy(), last=np.zeros((n, 4)))
            bad, newly, fac = batch._decide(sq, ids, a["rhos"], a["done"], a["last"], dims, tol, rtol, update_rho, 0, False,
                                            marked=marked)
            rho_c, last_c, fac_c = rhos0.copy(), np.zeros((n, 4)), np.zeros(n)
            status = np.zeros(n, dtype=np.int32)
            live = np.ascontiguousarray(~done0, dtype=np.uint8)
            mk = np.ascontiguousarray(marked, dtype=np.uint8)
            ev = lib.ggl_batch_decide(n, ptr(np.ascontiguousarray(sq)), live.ctypes.data_as(_lib._ubp),
                                      mk.ctypes.data_as(_lib._ubp), ptr(rho_c), ptr(dims), float(tol), float(rtol),
                                      int(update_rho), ptr(last_c), ptr(fac_c), status.ctypes.data_as(_lib._ip))
            assert ev == len(bad) + len(newly)
            assert np.array_equal(np.flatnonzero(status == 2), bad) and np.array_equal(np.flatnonzero(status == 1), newly)
            assert np.array_equal(rho_c, a["rhos"]) and np.array_equal(fac_c, fac)
            assert np.array_equal(last_c, a["last"])


def test_failed_points_say_why():
    """A point that ends as 'solver error' carries the reason -- the library's for the first marked instance of the point
    (HipEngine.failed_reason / ggl_failed_reason), or that its own sums stopped being finite -- and gets ONE warning."""
    import warnings
    from gglasso_amd import batch

    class Ctx:
        def failed_reason(self, k):
            return {5: "an eigensolver that did not converge (value 3.0)"}.get(k)

    assert batch._why(Ctx(), 4, 2) == "an eigensolver that did not converge (value 3.0)"       # instances 4, 5 of the point
    assert "not finite" in batch._why(Ctx(), 0, 2)                                              # nothing marked: the sums
    assert "not finite" in batch._why(object(), 0, 1)                                           # an engine without marks
    results = [({}, {'status': 'optimal'}), ({}, {'status': 'solver error', 'error': 'X'}), None,
               ({}, {'status': 'solver error'})]
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        batch._warn_failures(results)
    msgs = [str(w.message) for w in rec]
    assert len(msgs) == 2 and msgs[0].startswith("batch point 1: solver error -- X") and "batch point 3" in msgs[1]
    assert all(issubclass(w.category, RuntimeWarning) for w in rec)


def test_a_rebuilt_latent_component_that_differs_from_the_iterations_is_reported():
    """Host logic of batch._final_L (round 6): the L a batch returns is the one rebuilt from an eigendecomposition of the last
    L-step's input (solver/ggl_helper.py:29-36 is what the reference's callers apply numpy.linalg.matrix_rank to,
    helper/model_selection.py:638) -- and it IS the iteration's L up to the sign iteration's residual.  A rebuilt L that is
    something else (round 5's intermittent RANK table [[0,0,63],[0,0,108]] for [[160,78,6],[160,79,9]]) makes the point a
    'solver error' with NaN statistics, never a rank that reads like a result; a NaN L does the same."""
    from gglasso_amd import batch
    rng = np.random.default_rng(3)
    p = 30

    def low_rank(r):
        Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
        return (Q[:, :r] * rng.uniform(0.5, 2.0, r)) @ Q[:, :r].T

    L_it = [low_rank(4), low_rank(7), low_rank(2), low_rank(5)]
    noise = [1e-13 * rng.standard_normal((p, p)) for _ in range(4)]
    rebuilt = [L_it[0] + 0, np.zeros((p, p)), L_it[2] + 0, None]            # point 1: rebuilt from a lost input; point 3: not rebuilt

    class Eng:
        def finalize_L(self, which):
            assert which == 1
            return 3, np.array([4, 0, 2, -1], dtype=np.int32)

        def snapshot_L_k(self, i):
            return rebuilt[i].copy()

    sols = [{'L': L_it[k] + 0.5 * (noise[k] + noise[k].T)} for k in range(4)]
    sols[3]['L'] = L_it[3] + 0                                              # an eigendecomposition's L already: numpy's rule
    rk, bad = batch._final_L(Eng(), sols, 1)
    assert bad == [1] and rk.tolist() == [4, -1, 2, 5]
    assert np.array_equal(sols[0]['L'], rebuilt[0]) and np.array_equal(sols[2]['L'], rebuilt[2])      # the rebuilt L replaces
    assert np.abs(sols[1]['L'] - L_it[1]).max() < 1e-12                      # the doubtful one is left as the iteration had it
    results = [(sols[k], {'status': 'optimal'}) for k in range(4)]
    batch._mark_inconsistent(results, bad)
    assert [r[1]['status'] for r in results] == ['optimal', 'solver error', 'optimal', 'optimal']
    assert "rebuilt" in results[1][1]['error']
    # a NaN in a returned L (no rebuild): the point is a failed one, not an exception out of numpy
    sols2 = [{'L': np.full((p, p), np.nan)}]

    class Eng2:
        def finalize_L(self, which):
            return 0, np.array([-1], dtype=np.int32)

    rk2, bad2 = batch._final_L(Eng2(), sols2, 1)
    assert bad2 == [0] and rk2.tolist() == [-1]
    # an L that is exactly zero comes out of the sign iteration as noise relative to the step's INPUT, not to L (found by the
    # randomised sweep, tools/fuzz_parity.py 200 71 stats, case 142: 3.9e-12 at p = 16): consistent, and replaced by the zeros
    sols3 = [{'L': 4e-12 * (noise[0] + noise[0].T) / 1e-13, 'Theta': 30.0 * np.eye(p)}]

    class Eng3:
        def finalize_L(self, which):
            return 1, np.array([0], dtype=np.int32)

        def snapshot_L_k(self, i):
            return np.zeros((p, p))

    rk3, bad3 = batch._final_L(Eng3(), sols3, 1)
    assert bad3 == [] and rk3.tolist() == [0] and not sols3[0]['L'].any()
