#!/usr/bin/env python3
"""
Golden vectors for the model-selection row, generated like make_golden.py by
importing the REAL reference in the build container:

  G12 single_grid_search (helper/model_selection.py:505-692): a 5-point lambda1 grid, non-latent, and a
      3 x 2 (lambda1, mu1) grid with latent variables, at tol = rtol = 1e-10 so that the sequential warm-started
      walk of the reference and a batched identity-start solve agree far below the criteria's resolution.
      Stored: inputs (S, N, grids) and the reference's tables AIC, BIC[gamma], SP, RANK, the best point and
      the best Theta.
  G13 K_single_grid (:300-503): three data sets on a 3 x 2 latent grid; tables, ix_uniform / ix_indv / ix_mu and
      the uniformly and individually selected estimates.

    python tests/golden/make_golden_grid.py
"""
import numpy as np

import make_golden as mg


def main():
    mg._import_reference()
    from gglasso.helper import model_selection as ms
    from gglasso.helper import data_generation as dg
    p, N = 20, 150
    Sig, _ = dg.generate_precision_matrix(p=p, M=2, style='erdos', prob=0.25, seed=1240)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=1240)
    lam = np.logspace(-0.3, -1.5, 5)
    out = {"S": S, "N": np.array(N), "lambda_range": lam, "gamma": np.array(0.3)}
    best, est, _, st = mg.quiet(ms.single_grid_search, S, lam, N, method='eBIC', gamma=0.3, latent=False,
                                use_block=False, tol=1e-10, rtol=1e-10)
    gam = sorted(st['BIC'].keys())
    out["gammas"] = np.array(gam)
    out["plain_AIC"], out["plain_SP"] = st['AIC'], st['SP']
    out["plain_BIC"] = np.stack([st['BIC'][g] for g in gam])
    out["plain_best_lambda1"] = np.array(st['BEST']['lambda1'])
    out["plain_best_Theta"] = best['Theta']
    out["plain_estimates"] = est
    # AIC selection on the same grid
    best_a, _, _, st_a = mg.quiet(ms.single_grid_search, S, lam, N, method='AIC', gamma=0.3, latent=False,
                                  use_block=False, tol=1e-10, rtol=1e-10)
    out["plain_aic_best_lambda1"] = np.array(st_a['BEST']['lambda1'])
    out["plain_aic_best_Theta"] = best_a['Theta']

    lam2, mu = np.array([0.4, 0.2, 0.1]), np.array([1.0, 0.3])
    best, est, low, st = mg.quiet(ms.single_grid_search, S, lam2, N, method='eBIC', gamma=0.3, latent=True,
                                  mu_range=mu, use_block=False, tol=1e-10, rtol=1e-10)
    out["latent_lambda_range"], out["latent_mu_range"] = lam2, mu
    out["latent_AIC"], out["latent_SP"], out["latent_RANK"] = st['AIC'], st['SP'], st['RANK']
    out["latent_BIC"] = np.stack([st['BIC'][g] for g in gam])
    out["latent_best"] = np.array([st['BEST']['lambda1'], st['BEST']['mu1']])
    out["latent_best_Theta"], out["latent_best_L"] = best['Theta'], best['L']
    out["latent_lowrank"] = low
    mg.save("g12_single_grid_search", **out)

    # ---- G13 K_single_grid (model_selection.py:300-503): K = 3 data sets on a 3 x 2 (lambda1, mu1) grid, latent
    K = 3
    Sig, _ = dg.group_power_network(p, K=K, M=2, seed=1241)
    S3, _ = dg.sample_covariance_matrix(Sig, N, seed=1241)
    Nk = np.array([N, N + 30, N - 20])
    est_u, est_i, st = mg.quiet(ms.K_single_grid, S3, lam2, Nk, method='eBIC', gamma=0.3, latent=True, mu_range=mu,
                                use_block=False, store_all=True, tol=1e-10, rtol=1e-10)
    mg.save("g13_k_single_grid", S=S3, N=Nk, lambda_range=lam2, mu_range=mu, BIC=st['BIC'], AIC=st['AIC'], SP=st['SP'],
            RANK=st['RANK'], ix_uniform=np.array(st['ix_uniform']), ix_indv=st['ix_indv'], ix_mu=st['ix_mu'],
            uniform_Theta=est_u['Theta'], uniform_L=est_u['L'], indv_Theta=est_i['Theta'], indv_L=est_i['L'])


if __name__ == "__main__":
    main()
