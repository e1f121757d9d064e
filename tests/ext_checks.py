"""Shared assertions for ext_ADMM_MGL against the reference's vectors (fixtures G14 non-conforming, G15 conforming):
used with the oracle (CPU suite), with the product solver on the MI355X (-m gpu) and with the product's host loop over
the test-only oracle engine."""
import contextlib
import io

import numpy as np

NAMES = ("Omega", "Theta", "L", "X0", "X1")


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def g14_inputs(g):
    K = int(g["K"])
    S = {k: g[f"S_{k}"] for k in range(K)}
    p = g["p"]
    Om0 = {k: np.eye(p[k]) for k in range(K)}
    return K, p, S, g["G"], Om0


def _cmp(sol, g, tag, K, tol, names=NAMES):
    for nm in names:
        for k in range(K):
            err = np.abs(sol[nm][k] - g[f"{tag}_{nm}_{k}"]).max()
            assert err <= tol, (tag, nm, k, err)


def check_g14(load_golden, ext_ADMM_MGL, latent, measure_kw=True, traj_tol=1e-10):
    """ext_ADMM_MGL(S, lambda1, lambda2, reg, Omega_0, G, ...) with the reference's signature and return contract."""
    g = load_golden("g14_ext_admm_nonconforming")
    K, p, S, G, Om0 = g14_inputs(g)
    l1, l2, mu1 = (float(v) for v in g["params"])
    tag = "lat" if latent else "nol"
    kw = dict(latent=latent, mu1=mu1)
    for mi in (1, 2, 10):
        (sol, info), text = quiet(ext_ADMM_MGL, S, l1, l2, 'GGL', {k: v.copy() for k, v in Om0.items()}, G, max_iter=mi,
                                  tol=1e-20, rtol=1e-20, measure=True, **kw)
        _cmp(sol, g, f"{tag}_it{mi}", K, traj_tol)
        assert np.allclose(info['residual'], g[f"{tag}_it{mi}_residual"], rtol=1e-8)
        assert info['status'] == 'max iterations reached'
        assert f"ADMM terminated after {mi} iterations with status: max iterations reached." in text
        assert set(sol) >= set(NAMES) and sorted(sol['Theta']) == list(range(K))
        for k in range(K):
            assert sol['Theta'][k].shape == (p[k], p[k])
    (sol, info), _ = quiet(ext_ADMM_MGL, S, l1, l2, 'GGL', {k: v.copy() for k, v in Om0.items()}, G, tol=1e-9, rtol=1e-9,
                           measure=True, **kw)
    assert info['status'] == str(g[f"{tag}_conv_status"])
    assert abs(len(info['residual']) - int(g[f"{tag}_conv_iters"])) <= 1
    for k in range(K):
        assert np.linalg.norm(sol['Theta'][k] - g[f"{tag}_conv_Theta_{k}"]) <= 1e-8
    # warm start: both duals given, per-instance lambda1 / mu1, rho = 1.7
    Omw = {k: g[f"{tag}_warmstart_Omega_{k}"].copy() for k in range(K)}
    X0w = {k: g[f"{tag}_warmstart_X0_{k}"].copy() for k in range(K)}
    X1w = {k: g[f"{tag}_warmstart_X1_{k}"].copy() for k in range(K)}
    keep = [X0w[1].copy(), X1w[2].copy(), S[0].copy()]
    (sol, info), _ = quiet(ext_ADMM_MGL, S, g["warm_lambda1"], l2, 'GGL', Omw, G, X0=X0w, X1=X1w, rho=1.7, max_iter=5,
                           tol=1e-20, rtol=1e-20, measure=True, latent=latent, mu1=g["warm_mu1"])
    _cmp(sol, g, f"{tag}_warm", K, traj_tol)
    assert np.allclose(info['residual'], g[f"{tag}_warm_residual"], rtol=1e-8)
    assert np.array_equal(S[0], keep[2])          # the caller's S is never written
    return g


def check_g14_kkt(load_golden, ext_ADMM_MGL, latent):
    g = load_golden("g14_ext_admm_nonconforming")
    K, p, S, G, Om0 = g14_inputs(g)
    l1, l2, mu1 = (float(v) for v in g["params"])
    tag = "lat" if latent else "nol"
    (sol, info), _ = quiet(ext_ADMM_MGL, S, l1, l2, 'GGL', Om0, G, stopping_criterion='kkt', tol=1e-6, measure=True,
                           latent=latent, mu1=mu1)
    assert info['status'] == str(g[f"{tag}_kkt_status"])
    assert len(info['residual']) == len(g[f"{tag}_kkt_residual"])
    assert np.allclose(info['residual'], g[f"{tag}_kkt_residual"], rtol=1e-6, atol=1e-12)
    _cmp(sol, g, f"{tag}_kkt", K, 1e-9)


def check_g15(load_golden, ext_ADMM_MGL, latent):
    """Conforming variables, trivial G: the extended solver with lambda2/sqrt(K) solves ADMM_MGL's problem
    (reference tests/test_solvers.py:71-120, to 2 decimals there)."""
    g = load_golden("g15_ext_admm_conforming")
    S, G = g["S"], g["G"]
    K, p = S.shape[0], S.shape[1]
    l1, l2, mu1 = (float(v) for v in g["params"])
    tag = "lat" if latent else "nol"
    Sd = {k: S[k].copy() for k in range(K)}
    (sol, info), _ = quiet(ext_ADMM_MGL, Sd, l1, l2 / np.sqrt(K), 'GGL', {k: np.eye(p) for k in range(K)}, G, tol=1e-9,
                           rtol=1e-9, latent=latent, mu1=mu1, measure=True)
    assert info['status'] == str(g[f"{tag}_ext_status"])
    assert abs(len(info['residual']) - int(g[f"{tag}_ext_iters"])) <= 1
    Th = np.stack([sol['Theta'][k] for k in range(K)])
    assert np.linalg.norm(Th - g[f"{tag}_ext_Theta"]) <= 1e-8
    assert np.abs(Th - g[f"{tag}_mgl_Theta"]).max() <= 1e-2
    if latent:
        Ls = np.stack([sol['L'][k] for k in range(K)])
        assert np.abs(Ls - g[f"{tag}_ext_L"]).max() <= 1e-7
        assert np.abs(Ls - g[f"{tag}_mgl_L"]).max() <= 1e-2


def check_ext_batch(load_golden, ext_ADMM_MGL, ext_ADMM_MGL_batch, grid_search, latent):
    """ext_ADMM_MGL_batch: a 3 x 2 grid of (lambda1, lambda2) points of the non-conforming fixture problem (G14's S and G) as
    ONE batch -- every point must end where its own ext_ADMM_MGL call from the identity start ends (same status, same
    iteration count, same solution), and grid_search(solver=ext_ADMM_MGL, S dict, G) built on it must select what the
    reference's sequential warm-started walk with the same solver selects (helper/model_selection.py:208-224)."""
    g = load_golden("g14_ext_admm_nonconforming")
    K, p, S, G, Om0 = g14_inputs(g)
    l1s, l2s = [0.2, 0.08, 0.03], [0.1, 0.02]
    pts = [(a, b) for b in l2s for a in l1s]
    mu1 = 0.3 * np.ones(K) if latent else None
    res, _ = quiet(ext_ADMM_MGL_batch, S, [a for a, _ in pts], [b for _, b in pts], 'GGL', G, tol=1e-8, rtol=1e-8,
                   latent=latent, mu1=mu1)
    assert len(res) == len(pts)
    for (a, b), (sol, info) in zip(pts, res):
        (ref, rinfo), _ = quiet(ext_ADMM_MGL, S, a, b, 'GGL', {k: v.copy() for k, v in Om0.items()}, G, tol=1e-8, rtol=1e-8,
                                latent=latent, mu1=mu1, measure=True)
        assert info['status'] == rinfo['status'], (a, b)
        assert info['iterations'] == len(rinfo['residual']), (a, b, info['iterations'], len(rinfo['residual']))
        for nm in NAMES:
            for k in range(K):
                assert sol[nm][k].shape == ref[nm][k].shape
                assert np.abs(sol[nm][k] - ref[nm][k]).max() <= 1e-9, (a, b, nm, k)
    if grid_search is None:
        return
    N = 50 * np.ones(K)
    kw = dict(l1=np.array(l1s), l2=np.array(l2s), method='eBIC', gamma=0.3, G=G, tol=1e-8, rtol=1e-8)
    (stats_b, ix_b, best_b), _ = quiet(grid_search, ext_ADMM_MGL, S, N, p, 'GGL', **kw)
    (stats_s, ix_s, best_s), _ = quiet(grid_search, ext_ADMM_MGL, S, N, p, 'GGL', batched=False, **kw)
    # same selection, same selected estimate; the tables agree where the edge COUNT agrees (count_nonzero of an estimate
    # solved to 1e-8 from another start can differ by a few entries that sit at the threshold -- the reference's own table
    # depends on its warm-start order in the same way)
    assert tuple(ix_b) == tuple(ix_s)
    same_edges = np.isclose(stats_b['SP'], stats_s['SP'], atol=1e-12)
    assert same_edges.sum() >= same_edges.size - 2
    assert np.allclose(stats_b['BIC'][0.3][same_edges], stats_s['BIC'][0.3][same_edges], rtol=1e-6, atol=1e-4)
    assert np.allclose(stats_b['BIC'][0.3], stats_s['BIC'][0.3], rtol=5e-2)
    for k in range(K):
        assert np.abs(best_b["Theta"][k] - best_s["Theta"][k]).max() <= 1e-4      # two solves to dim*tol ~ 5e-6 from different starts
