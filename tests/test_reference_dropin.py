"""Drop-in check against the REAL reference package (build container only; skipped where /root/reference is
absent, e.g. on the GPU box).  The reference's own model-selection driver and problem class are run with
gglasso_amd's solvers plugged in at the two seams SURVEY.md section 8(b) names, and must produce what they produce
with their own solvers.  No GPU here, so the array work behind gglasso_amd's host loops is done by the test-only
oracle engine: what is verified is the boundary (signatures, kwargs, return dicts, status strings, warm starts),
which is exactly what a maintainer's one-line import swap relies on."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import pytest

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    shim = tempfile.mkdtemp(prefix="numba_shim_")
    os.makedirs(os.path.join(shim, "numba"))
    with open(os.path.join(shim, "numba", "__init__.py"), "w") as fh:
        fh.write("def _ident(*a, **k):\n    if len(a) == 1 and callable(a[0]) and not k:\n        return a[0]\n"
                 "    return lambda f: f\nnjit = jit = _ident\n")
    with open(os.path.join(shim, "numba", "typed.py"), "w") as fh:
        fh.write("List = list\n")
    sys.path.insert(0, REF_SRC)
    sys.path.insert(0, shim)
    import gglasso.problem as problem
    import gglasso.helper.model_selection as ms
    import gglasso.helper.data_generation as dg
    import gglasso.solver.admm_solver as admm
    import gglasso.solver.single_admm_solver as sadmm
    yield {"problem": problem, "ms": ms, "dg": dg, "admm": admm, "sadmm": sadmm}
    sys.path.remove(REF_SRC)
    sys.path.remove(shim)


@pytest.fixture()
def ours(monkeypatch):
    from gglasso_amd import solver
    from oracle_engine import OracleEngine
    monkeypatch.setattr(solver, "ENGINE", OracleEngine)
    return solver


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_seam1_grid_search_takes_our_solver(ref, ours):
    """helper/model_selection.py:55,222 -- grid_search(solver, ...) calls solver(**kwargs)."""
    dg, ms = ref["dg"], ref["ms"]
    K, p, N = 3, 20, 200
    Sig, _ = dg.group_power_network(p, K=K, M=2, seed=7)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=7)
    l1 = np.logspace(-1, -2, 3)
    l2 = np.logspace(-1, -2, 2)
    Nk = np.array([N] * K)
    a = quiet(ms.grid_search, ref["admm"].ADMM_MGL, S, Nk, p, 'GGL', l1, l2, method='eBIC', gamma=0.3,
              tol=1e-8, rtol=1e-8)
    b = quiet(ms.grid_search, ours.ADMM_MGL, S, Nk, p, 'GGL', l1, l2, method='eBIC', gamma=0.3, tol=1e-8, rtol=1e-8)
    stats_a, ix_a, sol_a = a
    stats_b, ix_b, sol_b = b
    assert ix_a == ix_b
    assert np.allclose(stats_a['BIC'][0.3], stats_b['BIC'][0.3], rtol=1e-8)
    assert np.allclose(stats_a['SP'], stats_b['SP'])
    assert np.abs(sol_a['Theta'] - sol_b['Theta']).max() <= 1e-7


def test_seam2_glasso_problem_with_patched_solvers(ref, ours, monkeypatch):
    """problem.py:10-11 imports ADMM_MGL / ADMM_SGL / block_SGL by name; swapping them is the whole integration."""
    dg, problem, ms = ref["dg"], ref["problem"], ref["ms"]
    K, p, N = 3, 20, 300
    Sig, _ = dg.time_varying_power_network(p, K=K, M=4, seed=9)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=9)

    def solve(reg, latent):
        P = problem.glasso_problem(S, N, reg=reg, reg_params={'lambda1': 0.05, 'lambda2': 0.02, 'mu1': 0.3},
                                   latent=latent, do_scaling=False)
        quiet(P.solve, tol=1e-9, rtol=1e-9)
        return P

    base = {(r, l): solve(r, l) for r in ('GGL', 'FGL') for l in (False, True)}
    monkeypatch.setattr(problem, "ADMM_MGL", ours.ADMM_MGL)
    monkeypatch.setattr(problem, "ADMM_SGL", ours.ADMM_SGL)
    monkeypatch.setattr(problem, "block_SGL", ours.block_SGL)
    monkeypatch.setattr(ms, "ADMM_SGL", ours.ADMM_SGL)
    monkeypatch.setattr(ms, "block_SGL", ours.block_SGL)
    for key, P0 in base.items():
        P1 = solve(*key)
        assert P1.solver_info['status'] == P0.solver_info['status'] == 'optimal'
        assert np.abs(P1.solution.precision_ - P0.solution.precision_).max() <= 1e-8
        if key[1]:
            assert np.abs(P1.solution.lowrank_ - P0.solution.lowrank_).max() <= 1e-8

    # single problem: non-latent goes through block_SGL (problem.py:443-450), latent through ADMM_SGL (:430-440)
    for latent in (False, True):
        monkeypatch.undo()
        P0 = problem.glasso_problem(S[0], N, reg=None, reg_params={'lambda1': 0.1, 'mu1': 0.4}, latent=latent,
                                    do_scaling=False)
        quiet(P0.solve, tol=1e-9, rtol=1e-9)
        from gglasso_amd import solver
        from oracle_engine import OracleEngine
        monkeypatch.setattr(solver, "ENGINE", OracleEngine)
        monkeypatch.setattr(problem, "ADMM_SGL", solver.ADMM_SGL)
        monkeypatch.setattr(problem, "block_SGL", solver.block_SGL)
        P1 = problem.glasso_problem(S[0], N, reg=None, reg_params={'lambda1': 0.1, 'mu1': 0.4}, latent=latent,
                                    do_scaling=False)
        quiet(P1.solve, tol=1e-9, rtol=1e-9)
        assert np.abs(P1.solution.precision_ - P0.solution.precision_).max() <= 1e-8


def test_seam2_glasso_problem_on_drawn_configurations(ref, monkeypatch):
    """The real glasso_problem class (problem.py) on sixty drawn configurations -- single / GGL / FGL, K = 2 .. 5, p = 6 .. 24, latent
    on / off, do_scaling on / off (the class scales S to correlations and the solution back, problem.py:239-262), drawn reg_params
    -- once with its own solvers, once with ours swapped in at the by-name imports (problem.py:10-11): same status, same
    precision_ (and lowrank_)."""
    dg, problem, ms = ref["dg"], ref["problem"], ref["ms"]
    from gglasso_amd import solver
    from oracle_engine import OracleEngine
    rng = np.random.default_rng(4242)
    for trial in range(60):
        single = bool(rng.random() < 0.35)
        K, p, N = int(rng.integers(2, 6)), int(rng.choice([6, 10, 16, 24])), int(rng.integers(60, 400))
        reg = None if single else ("GGL" if rng.random() < 0.5 else "FGL")
        latent, scaling = bool(rng.random() < 0.4), bool(rng.random() < 0.5)
        from gglasso_amd import synth
        Sk, _ = synth.make_problem(reg or "GGL", 1 if single else K, p, N=N, seed=int(rng.integers(1 << 30)))
        S = Sk[0] if single else Sk
        params = {'lambda1': float(10.0 ** rng.uniform(-1.6, -0.5)), 'lambda2': float(10.0 ** rng.uniform(-2.0, -0.7)),
                  'mu1': float(10.0 ** rng.uniform(-0.7, 0.3))}
        tag = dict(trial=trial, single=single, K=K, p=p, N=N, reg=reg, latent=latent, scaling=scaling, **params)

        def solve():
            P = problem.glasso_problem(S, N, reg=reg, reg_params=dict(params), latent=latent, do_scaling=scaling)
            quiet(P.solve, tol=1e-9, rtol=1e-9)
            return P

        monkeypatch.undo()
        P0 = solve()
        monkeypatch.setattr(solver, "ENGINE", OracleEngine)
        monkeypatch.setattr(problem, "ADMM_MGL", solver.ADMM_MGL)
        monkeypatch.setattr(problem, "ADMM_SGL", solver.ADMM_SGL)
        monkeypatch.setattr(problem, "block_SGL", solver.block_SGL)
        P1 = solve()
        # (the non-latent single problem goes through block_SGL, which returns no info: problem.py:443-450)
        assert P1.solver_info.get('status') == P0.solver_info.get('status'), tag
        assert np.abs(P1.solution.precision_ - P0.solution.precision_).max() <= 1e-8 * max(1.0, np.abs(P0.solution.precision_).max()), tag
        if latent:
            assert np.abs(P1.solution.lowrank_ - P0.solution.lowrank_).max() <= 1e-8 * max(1.0, np.abs(P0.solution.precision_).max()), tag


def test_seam2_model_selection_single(ref, ours, monkeypatch):
    """single_grid_search (model_selection.py:505) drives block_SGL / ADMM_SGL imported at :13."""
    dg, problem, ms = ref["dg"], ref["problem"], ref["ms"]
    p, N = 20, 200
    Sig, _ = dg.generate_precision_matrix(p=p, M=2, style='erdos', prob=0.2, seed=11)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=11)
    lam = np.logspace(0, -2, 5)
    a = quiet(ms.single_grid_search, S, lam, N, method='eBIC', gamma=0.3, latent=False, use_block=True,
              tol=1e-8, rtol=1e-8)
    monkeypatch.setattr(ms, "ADMM_SGL", ours.ADMM_SGL)
    monkeypatch.setattr(ms, "block_SGL", ours.block_SGL)
    b = quiet(ms.single_grid_search, S, lam, N, method='eBIC', gamma=0.3, latent=False, use_block=True,
              tol=1e-8, rtol=1e-8)
    assert np.abs(a[0]['Theta'] - b[0]['Theta']).max() <= 1e-7
    assert np.abs(a[1] - b[1]).max() <= 1e-7                       # estimates along the whole lambda path
    assert np.allclose(a[3]['BIC'][0.3], b[3]['BIC'][0.3], rtol=1e-7)
    assert a[3]['BEST'] == b[3]['BEST']


def test_seam3a_glasso_problem_model_selection_with_the_batched_grid(ref, ours, monkeypatch):
    """problem.py:14 imports single_grid_search by name and model_selection() calls it at :606; swapping in
    gglasso_amd.model_selection.single_grid_search (the whole grid as one batch) must select the same point and
    the same estimate as the reference's sequential walk, with and without latent variables."""
    dg, problem = ref["dg"], ref["problem"]
    from gglasso_amd import model_selection as ours_ms
    p, N = 20, 200
    Sig, _ = dg.generate_precision_matrix(p=p, M=2, style='erdos', prob=0.2, seed=13)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=13)

    def select(latent):
        P = problem.glasso_problem(S, N, reg=None, latent=latent, do_scaling=False)
        P.set_modelselect_params({'lambda1_range': np.logspace(0, -2, 5), 'mu1_range': np.array([1.0, 0.4])})
        quiet(P.model_selection, method='eBIC', gamma=0.3, tol=1e-9, rtol=1e-9)
        return P

    for latent in (False, True):
        monkeypatch.undo()
        P0 = select(latent)
        from gglasso_amd import solver
        from oracle_engine import OracleEngine
        monkeypatch.setattr(solver, "ENGINE", OracleEngine)
        monkeypatch.setattr(problem, "single_grid_search", ours_ms.single_grid_search)
        P1 = select(latent)
        assert P1.reg_params['lambda1'] == P0.reg_params['lambda1']
        if latent:
            assert P1.reg_params['mu1'] == P0.reg_params['mu1']
            assert np.abs(P1.solution.lowrank_ - P0.solution.lowrank_).max() <= 1e-6
        assert np.abs(P1.solution.precision_ - P0.solution.precision_).max() <= 1e-6
        assert np.allclose(P1.modelselect_stats['BIC'][0.3], P0.modelselect_stats['BIC'][0.3], rtol=1e-7)
        assert np.array_equal(P1.modelselect_stats['SP'], P0.modelselect_stats['SP'])
        assert np.array_equal(P1.modelselect_stats['RANK'], P0.modelselect_stats['RANK'])


def test_seam2_ext_admm_nonconforming_through_glasso_problem(ref, ours, monkeypatch):
    """problem.py:12,468 -- a dict S with a bookkeeping array G makes glasso_problem.solve() call ext_ADMM_MGL; swapping
    the name for gglasso_amd.ext_ADMM_MGL (instances padded into one stack) must give the reference's solution, with and
    without latent variables (reference tests/test_problem.py:117-138)."""
    import pandas as pd
    import gglasso.helper.ext_admm_helper as eh
    from gglasso_amd import ext_solver
    problem = ref["problem"]
    rng = np.random.default_rng(5)
    K, N = 4, 200
    all_obs, S = {}, {}
    for k in range(K):
        X = rng.random((5 + k, N))
        all_obs[k] = pd.DataFrame(X)
        S[k] = np.cov(X, bias=True)
    ix_exist, ix_location = eh.construct_indexer(list(all_obs.values()))
    G = quiet(eh.create_group_array, ix_exist, ix_location, 2)

    def solve(latent):
        P = problem.glasso_problem(S={k: v.copy() for k, v in S.items()}, N=N, reg="GGL", latent=latent, G=G,
                                   reg_params={'lambda1': 0.01, 'lambda2': 0.001, 'mu1': 0.05}, do_scaling=True)
        quiet(P.solve, tol=1e-9, rtol=1e-9)
        return P

    for latent in (False, True):
        monkeypatch.undo()
        P0 = solve(latent)
        from gglasso_amd import solver
        from oracle_engine import OracleEngine
        monkeypatch.setattr(solver, "ENGINE", OracleEngine)
        monkeypatch.setattr(problem, "ext_ADMM_MGL", ext_solver.ext_ADMM_MGL)
        P1 = solve(latent)
        assert P1.solver_info['status'] == P0.solver_info['status']
        for k in range(K):
            assert np.abs(P1.solution.precision_[k] - P0.solution.precision_[k]).max() <= 1e-8
            if latent:
                assert np.abs(P1.solution.lowrank_[k] - P0.solution.lowrank_[k]).max() <= 1e-8


def test_seam3b_glasso_problem_model_selection_mgl_with_the_batched_grid(ref, ours, monkeypatch):
    """problem.py:14,626-654: model_selection() of a multiple-graph problem calls K_single_grid (latent, stage one) and
    grid_search(solver=ADMM_MGL, ...); swapping in gglasso_amd's ADMM_MGL, grid_search and K_single_grid (every grid a
    batch) must select the same (lambda1, lambda2) and the same estimate as the reference's sequential walks."""
    dg, problem = ref["dg"], ref["problem"]
    from gglasso_amd import model_selection as ours_ms
    K, p, N = 3, 16, 200
    Sig, _ = dg.group_power_network(p, K=K, M=2, seed=17)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=17)

    def select(latent):
        P = problem.glasso_problem(S.copy(), N, reg='GGL', latent=latent, do_scaling=False)
        P.set_modelselect_params({'lambda1_range': np.logspace(-0.5, -1.5, 3), 'lambda2_range': np.array([0.1, 0.02]),
                                  'mu1_range': np.array([1.0, 0.4])})
        quiet(P.model_selection, method='eBIC', gamma=0.1, tol=1e-9, rtol=1e-9)
        return P

    for latent in (False, True):
        monkeypatch.undo()
        P0 = select(latent)
        from gglasso_amd import solver
        from oracle_engine import OracleEngine
        monkeypatch.setattr(solver, "ENGINE", OracleEngine)
        monkeypatch.setattr(problem, "ADMM_MGL", solver.ADMM_MGL)
        monkeypatch.setattr(problem, "grid_search", ours_ms.grid_search)
        monkeypatch.setattr(problem, "K_single_grid", ours_ms.K_single_grid)
        P1 = select(latent)
        assert P1.reg_params['lambda1'] == P0.reg_params['lambda1'] and P1.reg_params['lambda2'] == P0.reg_params['lambda2']
        assert np.abs(P1.solution.precision_ - P0.solution.precision_).max() <= 1e-6
        assert np.allclose(P1.modelselect_stats['BIC'][0.1], P0.modelselect_stats['BIC'][0.1], rtol=1e-7)
        assert np.array_equal(P1.modelselect_stats['SP'], P0.modelselect_stats['SP'])
        if latent:
            assert np.abs(P1.solution.lowrank_ - P0.solution.lowrank_).max() <= 1e-6
            assert np.array_equal(P1.modelselect_stats['RANK'], P0.modelselect_stats['RANK'])


def test_seam1_reference_grid_search_rank_table_above_the_jacobi_limit(ref, ours):
    """the REFERENCE's grid_search (helper/model_selection.py:55-298) applies numpy.linalg.matrix_rank to the
    sol['L'] our solver hands it (:254).  Driven by gglasso_amd.ADMM_MGL at p = 160 it must reproduce the RANK table the
    reference computed with its own solver (fixture G18).  (Array work by the oracle engine here; tests/test_gpu_latent_rank.py
    runs the same walk on the HIP engine, where the returned L is the rebuilt one.)"""
    from conftest import load_golden
    g = load_golden("g18_latent_rank_large_p")
    ms = ref["ms"]
    S, Nk = g["mgl_S"], g["grid_N"]
    stats, ix, best = quiet(ms.grid_search, ours.ADMM_MGL, S, Nk, S.shape[1], "GGL", g["grid_l1"], l2=g["grid_l2"],
                            method='eBIC', gamma=0.3, latent=True, mu_range=g["grid_mu_range"], ix_mu=g["grid_ix_mu"],
                            tol=1e-10, rtol=1e-10)
    assert np.array_equal(stats['RANK'], g["grid_RANK"])
    assert tuple(int(v) for v in ix) == tuple(int(v) for v in g["grid_ix"])
    assert np.allclose(stats['SP'], g["grid_SP"])
    assert [np.linalg.matrix_rank(best['L'][k]) for k in range(S.shape[0])] == list(g["grid_best_rank"])
