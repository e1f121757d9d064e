"""ext_ADMM_MGL on the MI355X (ggl_ext_* entry points, csrc/ext_group.hip) against the reference's vectors G14 / G15
and, at sizes beyond the fixtures, against the CPU oracle: instances of different dimension in one padded stack."""
import numpy as np
import pytest

import ext_checks
from conftest import load_golden
from oracle import ggl_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    from gglasso_amd import ext_solver
    return ext_solver.ext_ADMM_MGL


@pytest.mark.parametrize("latent", [False, True])
def test_g14_nonconforming_trajectories(ext, latent):
    ext_checks.check_g14(load_golden, ext, latent)


@pytest.mark.parametrize("latent", [False, True])
def test_g14_kkt_criterion(ext, latent):
    ext_checks.check_g14_kkt(load_golden, ext, latent)


@pytest.mark.parametrize("latent", [False, True])
def test_ext_batched_grid(ext, latent):
    """The (lambda1, lambda2) grid of an ext_ADMM_MGL problem as ONE batch on the GPU (ggl_ext_setup_batch /
    ggl_ext_batch_step: group shrink with a grid-point dimension, per-problem sums), every point against its own
    ext_ADMM_MGL solve; grid_search on top of it against the sequential walk."""
    from gglasso_amd import ext_solver, model_selection
    ext_checks.check_ext_batch(load_golden, ext, ext_solver.ext_ADMM_MGL_batch,
                               None if latent else model_selection.grid_search, latent)


@pytest.mark.parametrize("latent", [False, True])
def test_g15_conforming_equals_admm_mgl(ext, latent):
    """reference tests/test_solvers.py:71-120: trivial G with lambda2/sqrt(K) solves ADMM_MGL's problem."""
    ext_checks.check_g15(load_golden, ext, latent)
    g = load_golden("g15_ext_admm_conforming")
    from gglasso_amd import solver
    S = g["S"]
    K, p = S.shape[:2]
    l1, l2, mu1 = (float(v) for v in g["params"])
    tag = "lat" if latent else "nol"
    (sol, _), _ = ext_checks.quiet(solver.ADMM_MGL, S, l1, l2, 'GGL', np.stack([np.eye(p)] * K), tol=1e-9, rtol=1e-9,
                                   latent=latent, mu1=mu1)
    assert np.linalg.norm(sol['Theta'] - g[f"{tag}_mgl_Theta"]) <= 1e-8


def _nonconforming(rng, universe, sizes, prob=0.08):
    """Instances observing random subsets of a common variable set; G like helper/ext_admm_helper.py:104-144
    (one group per pair of variables present together in >= 2 instances)."""
    members = [np.sort(rng.choice(universe, size=n, replace=False)) for n in sizes]
    A = rng.standard_normal((universe, universe)) * (rng.random((universe, universe)) < prob)
    Prec = A @ A.T + np.eye(universe)
    Sig = np.linalg.inv(Prec)
    S = {}
    for k, ix in enumerate(members):
        X = rng.multivariate_normal(np.zeros(universe), Sig, size=3 * len(ix)).T[ix]
        S[k] = np.cov(X, bias=True)
    K = len(sizes)
    loc = -np.ones((universe, K), dtype=int)
    for k, ix in enumerate(members):
        loc[ix, k] = np.arange(len(ix))
    rows = []
    for a in range(universe):
        for b in range(a + 1, universe):
            both = (loc[a] >= 0) & (loc[b] >= 0)
            if both.sum() >= 2:
                g0 = np.where(both, np.minimum(loc[a], loc[b]), -1)
                g1 = np.where(both, np.maximum(loc[a], loc[b]), -1)
                rows.append((g0, g1))
    G = np.stack([np.stack([r[0] for r in rows]), np.stack([r[1] for r in rows])]).astype(int)
    return S, G, np.array([len(ix) for ix in members])


@pytest.mark.parametrize("sizes,latent", [((140, 170, 150, 160, 128), False), ((150, 180, 165), True), ((40, 64, 50, 33), True)])
def test_padded_stack_vs_oracle_mid_sizes(ext, sizes, latent):
    """p_k on both sides of the LDS-Jacobi limit (128): Newton-Schulz Omega-/L-steps on block-diagonal padded
    matrices, speculation included; 12 iterations and a converged solve against the oracle."""
    rng = np.random.default_rng(sum(sizes))
    S, G, p = _nonconforming(rng, max(sizes) + 20, sizes)
    orc.check_G(G, p)
    K = len(sizes)
    Om0 = {k: np.eye(p[k]) for k in range(K)}
    lam1 = np.linspace(0.04, 0.07, K)
    kw = dict(latent=latent, mu1=0.3 * np.ones(K))
    ref, rinfo = orc.ext_ADMM_MGL(S, lam1, 0.03, 'GGL', Om0, G, max_iter=12, tol=1e-20, rtol=1e-20, measure=True, **kw)
    (sol, info), _ = ext_checks.quiet(ext, S, lam1, 0.03, 'GGL', Om0, G, max_iter=12, tol=1e-20, rtol=1e-20, measure=True,
                                      **kw)
    for nm in ext_checks.NAMES:
        for k in range(K):
            assert np.abs(sol[nm][k] - ref[nm][k]).max() <= 1e-9, (nm, k)
    assert np.allclose(info['residual'], rinfo['residual'], rtol=1e-8)
    # (looser tolerances than the fixtures' 1e-9: the oracle walks the instances in Python, ~0.1 s per iteration here)
    ref, rinfo = orc.ext_ADMM_MGL(S, lam1, 0.03, 'GGL', Om0, G, tol=1e-6, rtol=1e-5, **kw)
    (sol, info), _ = ext_checks.quiet(ext, S, lam1, 0.03, 'GGL', Om0, G, tol=1e-6, rtol=1e-5, measure=True, **kw)
    assert info['status'] == rinfo['status'] and len(info['residual']) == rinfo['iterations']
    for k in range(K):
        assert np.linalg.norm(sol['Theta'][k] - ref['Theta'][k]) <= 1e-8
        assert np.array_equal(sol['Theta'][k], sol['Theta'][k].T)


def test_group_shrink_kernel_against_reference_vectors():
    """prox_2norm_G on its own through the KKT entry point's building blocks is covered by the KKT test; here the
    gather/scatter kernel is isolated: one ext step from a state where everything else is the identity map."""
    from gglasso_amd import solver
    g = load_golden("g14_ext_admm_nonconforming")
    K, p, S, G, _ = ext_checks.g14_inputs(g)
    P = int(p.max())
    from gglasso_amd.ext_solver import _pad
    Z = {k: g[f"proxG_in_{k}"] for k in range(K)}
    # Theta-step with a huge rho: V = (Omega + X0 + Lambda - X1)/2 = Z when Omega = Lambda = Z, duals zero, and the
    # threshold lambda1/(2 rho) -> 0; Lambda_new = prox_2norm_G(Theta + X1, lambda2/rho)
    lam = float(g["proxG1_lam"])
    rho = 1e6
    eng = solver.HipEngine(_pad(S, K, p, P, True), _pad(Z, K, p, P, True), _pad(Z, K, p, P, True), np.zeros((K, P, P)),
                           eig=1)
    try:
        eng.ext_setup(p, G)
        eng.ext_set_state(_pad(Z, K, p, P, True), None)
        # skip the Omega-step's effect: call the pieces directly is not exposed, so compare after one full step on
        # Lambda only -- Omega changes, Theta = (Omega + Z)/2 changes; instead verify the identity through the oracle
        sq = eng.ext_step(rho, np.full(K, 1e-30), lam * rho, False, None).copy()
        st, xs = eng.state(), eng.ext_state()
    finally:
        eng.close()
    Th = {k: st['Theta'][k, :p[k], :p[k]] for k in range(K)}
    want = orc.prox_2norm_G(Th, G, lam)
    for k in range(K):
        assert np.abs(xs['Lambda'][k, :p[k], :p[k]] - want[k]).max() <= 1e-14, k
        assert np.array_equal(xs['Lambda'][k], xs['Lambda'][k].T)
    assert np.isfinite(sq).all()


def test_duplicate_group_entries_are_rejected(ext):
    g = load_golden("g14_ext_admm_nonconforming")
    K, p, S, G, Om0 = ext_checks.g14_inputs(g)
    Gd = np.concatenate([G, G[:, :1, :]], axis=1)         # first group listed twice
    with pytest.raises(AssertionError, match="more than one group"):
        ext(S, 0.1, 0.1, 'GGL', Om0, Gd)
