#!/usr/bin/env python3
"""
Golden vectors for the lambda1 x lambda2 model-selection grid of the multiple-graph problems (SURVEY.md section 8f
rank 1, second half), generated like make_golden.py by importing the REAL reference in the build container:

  G16 grid_search (helper/model_selection.py:55-298) with solver = the reference's ADMM_MGL on a 3 x 2 grid
      (l1 = 3 values, l2 = 2 values): GGL and FGL without latent variables, GGL with latent variables (mu_range and
      ix_mu given), and GGL with thresholding=True; tol = rtol = 1e-10 so that the warm-started sequential walk of the
      reference and a batched identity-start solve agree far below the criteria's resolution.  Stored: S, N, grids and
      the reference's AIC, BIC[gamma], SP, RANK, TAU tables, the selected index, BEST and the best solution's Theta / L;
      for thresholding also the un-thresholded best point.
  G17 single_grid_search with thresholding=True (tune_threshold, :698-737) on the G12 problem.

    python tests/golden/make_golden_mgl_grid.py
"""
import numpy as np

import make_golden as mg


def _store(out, tag, stats, ix, best, gam, latent=False):
    out[f"{tag}_AIC"], out[f"{tag}_SP"], out[f"{tag}_RANK"] = stats['AIC'], stats['SP'], stats['RANK']
    out[f"{tag}_BIC"] = np.stack([stats['BIC'][g] for g in gam])
    out[f"{tag}_ix"] = np.array(ix)
    out[f"{tag}_best"] = np.array([stats['BEST']['lambda1'], stats['BEST']['lambda2']])
    out[f"{tag}_best_Theta"] = best['Theta']
    if latent:
        out[f"{tag}_best_L"] = best['L']


def main():
    mg._import_reference()
    from gglasso.helper import model_selection as ms
    from gglasso.helper import data_generation as dg
    from gglasso.solver.admm_solver import ADMM_MGL
    p, K, N = 18, 3, 120
    l1 = np.array([0.25, 0.12, 0.06])
    l2 = np.array([0.08, 0.02])
    out = {"l1": l1, "l2": l2, "gamma": np.array(0.3)}
    Nk = np.array([N, N + 20, N - 10])
    out["N"] = Nk
    gam = None
    for reg in ("GGL", "FGL"):
        if reg == "GGL":
            Sig, _ = dg.group_power_network(p, K=K, M=3, seed=1250)
        else:
            Sig, _ = dg.time_varying_power_network(p, K=K, M=3, seed=1250)
        S, _ = dg.sample_covariance_matrix(Sig, N, seed=1250)
        out[f"S_{reg}"] = S
        stats, ix, best = mg.quiet(ms.grid_search, ADMM_MGL, S, Nk, p, reg, l1, l2=l2, method='eBIC', gamma=0.3,
                                   tol=1e-10, rtol=1e-10)
        gam = sorted(stats['BIC'].keys())
        _store(out, f"{reg}_plain", stats, ix, best, gam)
        out["L1"], out["L2"] = stats['L1'], stats['L2']
    out["gammas"] = np.array(gam)
    S = out["S_GGL"]
    # AIC selection
    stats, ix, best = mg.quiet(ms.grid_search, ADMM_MGL, S, Nk, p, "GGL", l1, l2=l2, method='AIC', gamma=0.3, tol=1e-10,
                               rtol=1e-10)
    _store(out, "GGL_aic", stats, ix, best, gam)
    # w2 parametrisation of the grid (:20-52)
    w2 = np.array([0.3, 0.1])
    stats, ix, best = mg.quiet(ms.grid_search, ADMM_MGL, S, Nk, p, "GGL", l1, w2=w2, method='eBIC', gamma=0.3, tol=1e-10,
                               rtol=1e-10)
    _store(out, "GGL_w2", stats, ix, best, gam)
    out["w2"], out["w2_L1"], out["w2_L2"] = w2, stats['L1'], stats['L2']
    # latent: mu chosen per instance and per lambda1 column through ix_mu (:216-220)
    mu_range = np.array([0.8, 0.3])
    ix_mu = np.array([[0, 1, 1], [1, 1, 0], [0, 0, 1]])
    stats, ix, best = mg.quiet(ms.grid_search, ADMM_MGL, S, Nk, p, "GGL", l1, l2=l2, method='eBIC', gamma=0.3, latent=True,
                               mu_range=mu_range, ix_mu=ix_mu, tol=1e-10, rtol=1e-10)
    _store(out, "GGL_latent", stats, ix, best, gam, latent=True)
    out["mu_range"], out["ix_mu"] = mu_range, ix_mu
    # thresholding (:226-247)
    stats, ix, best = mg.quiet(ms.grid_search, ADMM_MGL, S, Nk, p, "GGL", l1, l2=l2, method='eBIC', gamma=0.3,
                               thresholding=True, tol=1e-10, rtol=1e-10)
    _store(out, "GGL_thr", stats, ix, best, gam)
    out["GGL_thr_TAU"] = stats['TAU']
    out["GGL_thr_nothr_best"] = np.array([stats['NO_THRESHOLDING_BEST']['lambda1'], stats['NO_THRESHOLDING_BEST']['lambda2']])
    out["GGL_thr_nothr_Theta"] = stats['NO_THRESHOLDING_SOL']['Theta']
    mg.save("g16_mgl_grid_search", **out)

    g12 = np.load(mg.os.path.join(mg.HERE, "g12_single_grid_search.npz"))
    S1, N1, lam = g12["S"], int(g12["N"]), g12["lambda_range"]
    best, est, _, st = mg.quiet(ms.single_grid_search, S1, lam, N1, method='eBIC', gamma=0.3, latent=False,
                                thresholding=True, use_block=False, tol=1e-10, rtol=1e-10)
    gam1 = sorted(st['BIC'].keys())
    mg.save("g17_single_grid_thresholding", AIC=st['AIC'], BIC=np.stack([st['BIC'][g] for g in gam1]), SP=st['SP'],
            TAU=st['TAU'], best_lambda1=np.array(st['BEST']['lambda1']), best_Theta=best['Theta'], estimates=est)


if __name__ == "__main__":
    main()
