"""Shared assertions for the batched single_grid_search against the reference's tables (fixture G12)."""
import numpy as np


def check_single_grid_search(load_golden):
    from gglasso_amd import model_selection as ms
    g = load_golden("g12_single_grid_search")
    S, N, lam = g["S"], int(g["N"]), g["lambda_range"]
    gam = [float(x) for x in g["gammas"]]
    best, est, low, st = ms.single_grid_search(S, lam, N, method='eBIC', gamma=0.3, tol=1e-10, rtol=1e-10)
    assert sorted(st['BIC'].keys()) == gam and st['GAMMA'] == gam
    assert st['AIC'].shape == (len(lam), 1) and est.shape == (len(lam), 1) + S.shape
    # the fit term N(<S,Theta> - logdet Theta) is ~1e3: agreement to 1e-6 relative is far below one edge's weight
    assert np.allclose(st['AIC'], g["plain_AIC"], rtol=1e-7, atol=1e-5)
    for i, gm in enumerate(gam):
        assert np.allclose(st['BIC'][gm], g["plain_BIC"][i], rtol=1e-7, atol=1e-5), gm
    assert np.array_equal(st['SP'], g["plain_SP"])              # identical zero patterns at every grid point
    assert np.abs(est - g["plain_estimates"]).max() <= 1e-7
    assert float(st['BEST']['lambda1']) == float(g["plain_best_lambda1"]) and float(st['BEST']['mu1']) == 0.0
    # both walks stop at r <= dim*tol = 2.1e-8: a warm-started and an identity-started solve agree to that, not better
    assert np.linalg.norm(best['Theta'] - g["plain_best_Theta"]) <= 2e-7
    assert set(best) == {'Omega', 'Theta', 'X'} and low is not None and not low.any()
    best_a, _, _, st_a = ms.single_grid_search(S, lam, N, method='AIC', gamma=0.3, tol=1e-10, rtol=1e-10,
                                               store_all=False)
    assert float(st_a['BEST']['lambda1']) == float(g["plain_aic_best_lambda1"])
    assert np.linalg.norm(best_a['Theta'] - g["plain_aic_best_Theta"]) <= 2e-7

    lam2, mu = g["latent_lambda_range"], g["latent_mu_range"]
    best, est, low, st = ms.single_grid_search(S, lam2, N, method='eBIC', gamma=0.3, latent=True, mu_range=mu,
                                               tol=1e-10, rtol=1e-10)
    assert st['AIC'].shape == (3, 2)
    assert np.allclose(st['AIC'], g["latent_AIC"], rtol=1e-7, atol=1e-5)
    for i, gm in enumerate(gam):
        assert np.allclose(st['BIC'][gm], g["latent_BIC"][i], rtol=1e-7, atol=1e-5), gm
    assert np.array_equal(st['SP'], g["latent_SP"]) and np.array_equal(st['RANK'], g["latent_RANK"])
    assert [float(st['BEST']['lambda1']), float(st['BEST']['mu1'])] == [float(x) for x in g["latent_best"]]
    assert np.linalg.norm(best['Theta'] - g["latent_best_Theta"]) <= 2e-7
    assert np.linalg.norm(best['L'] - g["latent_best_L"]) <= 2e-7
    assert np.abs(low - g["latent_lowrank"]).max() <= 2e-7
    assert np.array_equal(st['LAMBDA'], np.meshgrid(mu, lam2)[1]) and np.array_equal(st['MU'], np.meshgrid(mu, lam2)[0])
    # a lambda1_mask grid runs point by point with the reference's warm start and host-side criteria: with a mask
    # of ones it must reproduce the batch's tables
    _, est_m, _, st_m = ms.single_grid_search(S, lam, N, method='eBIC', gamma=0.3, tol=1e-10, rtol=1e-10,
                                              lambda1_mask=np.ones_like(S))
    assert np.allclose(st_m['AIC'], g["plain_AIC"], rtol=1e-7, atol=1e-5)
    assert np.allclose(st_m['BIC'][0.3], g["plain_BIC"][gam.index(0.3)], rtol=1e-7, atol=1e-5)
    assert np.array_equal(st_m['SP'], g["plain_SP"]) and np.abs(est_m - g["plain_estimates"]).max() <= 1e-7
    # thresholded estimators (tune_threshold, model_selection.py:698-737) against the reference's tables, fixture G17
    t = load_golden("g17_single_grid_thresholding")
    best_t, est_t, _, st_t = ms.single_grid_search(S, lam, N, method='eBIC', gamma=0.3, thresholding=True, tol=1e-10,
                                                   rtol=1e-10)
    assert np.array_equal(st_t['TAU'], t["TAU"]) and np.array_equal(st_t['SP'], t["SP"])
    assert np.allclose(st_t['AIC'], t["AIC"], rtol=1e-7, atol=1e-5)
    for i, gm in enumerate(gam):
        assert np.allclose(st_t['BIC'][gm], t["BIC"][i], rtol=1e-7, atol=1e-5), gm
    assert float(st_t['BEST']['lambda1']) == float(t["best_lambda1"])
    assert np.abs(est_t - t["estimates"]).max() <= 1e-7 and np.linalg.norm(best_t['Theta'] - t["best_Theta"]) <= 2e-7


def check_k_single_grid(load_golden):
    """K data sets x (lambda1, mu1) grid as one batch against the reference's K_single_grid (fixture G13)."""
    from gglasso_amd import model_selection as ms
    g = load_golden("g13_k_single_grid")
    S, N, lam, mu = g["S"], g["N"], g["lambda_range"], g["mu_range"]
    est_u, est_i, st = ms.K_single_grid(S, lam, N, method='eBIC', gamma=0.3, latent=True, mu_range=mu, tol=1e-10,
                                        rtol=1e-10)
    assert st['BIC'].shape == (3, 3, 2)
    assert np.allclose(st['BIC'], g["BIC"], rtol=1e-7, atol=1e-5) and np.allclose(st['AIC'], g["AIC"], rtol=1e-7, atol=1e-5)
    assert np.array_equal(st['SP'], g["SP"]) and np.array_equal(st['RANK'], g["RANK"])
    assert int(st['ix_uniform']) == int(g["ix_uniform"])
    assert np.array_equal(st['ix_indv'], g["ix_indv"]) and np.array_equal(st['ix_mu'], g["ix_mu"])
    for ours, ref in ((est_u['Theta'], g["uniform_Theta"]), (est_u['L'], g["uniform_L"]),
                      (est_i['Theta'], g["indv_Theta"]), (est_i['L'], g["indv_L"])):
        assert ours.shape == ref.shape and np.abs(ours - ref).max() <= 2e-7
    # two small batches instead of one must give the same tables
    _, _, st2 = ms.K_single_grid(S, lam, N, method='eBIC', gamma=0.3, latent=True, mu_range=mu, tol=1e-10, rtol=1e-10,
                                 store_all=False, max_batch_bytes=14 * 6 * S.shape[1] ** 2 * 8 * 2)
    assert np.allclose(st2['BIC'], st['BIC'], rtol=1e-9) and np.array_equal(st2['ix_mu'], st['ix_mu'])


def check_mgl_grid_search(load_golden, group=None, tags=("GGL_plain", "FGL_plain", "GGL_aic", "GGL_w2", "GGL_latent", "GGL_thr")):
    """The lambda1 x lambda2 grid of the multiple-graph problems as one batch against the reference's grid_search
    tables (fixture G16: 3 x 2 grid; GGL / FGL, AIC selection, w2 parametrisation, latent with ix_mu, thresholding)."""
    from gglasso_amd import model_selection as ms
    from gglasso_amd.solver import ADMM_MGL
    g = load_golden("g16_mgl_grid_search")
    l1, l2, N = g["l1"], g["l2"], g["N"]
    gam = [float(x) for x in g["gammas"]]
    for tag in tags:
        reg = tag.split("_")[0]
        S = g[f"S_{reg}"]
        p = S.shape[1]
        kw = dict(l2=l2, method='eBIC', gamma=0.3, tol=1e-10, rtol=1e-10, group=group)
        if tag == "GGL_aic":
            kw["method"] = 'AIC'
        if tag == "GGL_w2":
            kw.pop("l2")
            kw["w2"] = g["w2"]
        if tag == "GGL_latent":
            kw.update(latent=True, mu_range=g["mu_range"], ix_mu=g["ix_mu"])
        if tag == "GGL_thr":
            kw["thresholding"] = True
        stats, ix, best = ms.grid_search(ADMM_MGL, S, N, p, reg, l1, **kw)
        assert stats['AIC'].shape == (2, 3) and sorted(stats['BIC']) == gam and stats['GAMMA'] == gam
        if tag == "GGL_w2":
            assert np.allclose(stats['L1'], g["w2_L1"]) and np.allclose(stats['L2'], g["w2_L2"])
        else:
            assert np.array_equal(stats['L1'], g["L1"]) and np.array_equal(stats['L2'], g["L2"])
        # the fit term N(<S,Theta> - logdet Theta) is ~1e3 per instance; one edge weighs >= 1
        assert np.allclose(stats['AIC'], g[f"{tag}_AIC"], rtol=1e-7, atol=1e-4), tag
        for i, gm in enumerate(gam):
            assert np.allclose(stats['BIC'][gm], g[f"{tag}_BIC"][i], rtol=1e-7, atol=1e-4), (tag, gm)
        assert np.array_equal(stats['SP'], g[f"{tag}_SP"]), tag               # identical zero patterns everywhere
        assert tuple(int(v) for v in ix) == tuple(int(v) for v in g[f"{tag}_ix"]), tag
        assert [float(stats['BEST']['lambda1']), float(stats['BEST']['lambda2'])] == [float(v) for v in g[f"{tag}_best"]]
        # both walks stop at r <= dim*tol: identity-started and warm-started solves agree to that, not better
        assert np.linalg.norm(best['Theta'] - g[f"{tag}_best_Theta"]) <= 5e-7, tag
        if tag == "GGL_latent":
            assert np.array_equal(stats['RANK'], g[f"{tag}_RANK"])
            assert np.linalg.norm(best['L'] - g[f"{tag}_best_L"]) <= 5e-7
        else:
            assert np.isnan(stats['RANK']).all()
        if tag == "GGL_thr":
            assert np.array_equal(stats['TAU'], g["GGL_thr_TAU"])
            nb = stats['NO_THRESHOLDING_BEST']
            assert [float(nb['lambda1']), float(nb['lambda2'])] == [float(v) for v in g["GGL_thr_nothr_best"]]
            assert np.linalg.norm(stats['NO_THRESHOLDING_SOL']['Theta'] - g["GGL_thr_nothr_Theta"]) <= 5e-7
        else:
            assert stats['TAU'] is None
    # the sequential walk with a solver callable (the reference's mode) gives the same tables as the batch
    S = g["S_GGL"]
    stats_s, ix_s, best_s = ms.grid_search(ADMM_MGL, S, N, S.shape[1], "GGL", l1, l2=l2, tol=1e-10, rtol=1e-10,
                                           batched=False)
    assert np.allclose(stats_s['AIC'], g["GGL_plain_AIC"], rtol=1e-7, atol=1e-4)
    assert tuple(int(v) for v in ix_s) == tuple(int(v) for v in g["GGL_plain_ix"])
    # a one-row grid (the reference's squeeze breaks on it) is served
    stats_1, ix_1, _ = ms.grid_search(ADMM_MGL, S, N, S.shape[1], "GGL", l1, l2=l2[:1], tol=1e-10, rtol=1e-10)
    assert stats_1['AIC'].shape == (1, 3) and np.allclose(stats_1['AIC'][0], g["GGL_plain_AIC"][0], rtol=1e-7, atol=1e-4)
