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
    try:
        ms.single_grid_search(S, lam, N, thresholding=True)
    except NotImplementedError:
        pass
    else:
        raise AssertionError("thresholding must be refused, not silently ignored")


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
