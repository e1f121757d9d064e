#!/usr/bin/env python3
"""
G18: the latent component ABOVE the LDS-Jacobi limit (p > 128), generated like make_golden.py by importing the REAL
reference in the build container.  Every earlier latent fixture has p <= 128, where this library's L-step is an
eigendecomposition; above it the per-iteration L-step is a sign iteration and the returned L is rebuilt from one
eigendecomposition at the end of the solve (ggl_finalize_L) -- what these vectors pin is the PROPERTY the reference's
callers rely on, numpy.linalg.matrix_rank(sol['L']) (helper/model_selection.py:254, :638), besides the entries.

  sgl_*    ADMM_SGL (solver/single_admm_solver.py:15-275), p = 200, latent, lambda1 = 0.03, four mu1: the reference's
           matrix_rank(L) for each, Theta and L in full for mu1 = 0.8
  sgs_*    single_grid_search (helper/model_selection.py:505-692) on the same S: 2 x 3 (lambda1, mu1) grid, RANK / SP /
           AIC / BIC tables and the chosen point
  mgl_*    ADMM_MGL (solver/admm_solver.py:13-313), K = 2, p = 160, latent, GGL (Theta, L in full) and FGL (L in full):
           matrix_rank(L_k)
  grid_*   grid_search (helper/model_selection.py:55-298) with solver = the reference's ADMM_MGL on that S: 2 x 2
           (lambda1, lambda2) grid, latent with mu_range / ix_mu: the RANK table (K, 2, 2) and the chosen point

The data: the observed marginal of a Gaussian with h = 5 hidden variables tied to most observed ones (sparse observed
precision minus a rank-5 term), so that the latent component has a rank that moves with mu1 (0 .. 8 here).

    python tests/golden/make_golden_rank.py
"""
import numpy as np

import make_golden as mg


def latent_cov(dg, p, h, seed, K=1, N=None):
    rng = np.random.default_rng(seed)
    _, Th_o = dg.generate_precision_matrix(p=p, M=p // 20, style='erdos', prob=0.15, seed=seed)
    out = []
    for k in range(K):
        B = rng.standard_normal((p, h)) * 0.12 * (rng.random((p, h)) < 0.7)
        marg = Th_o + 0.3 * k * np.eye(p) - B @ B.T
        marg += max(0.0, 0.5 - np.linalg.eigvalsh(marg).min()) * np.eye(p)      # smallest eigenvalue of the marginal precision: 0.5
        out.append(np.linalg.inv(marg))
    S, _ = dg.sample_covariance_matrix(np.stack(out), N or 4 * p, seed=seed)
    return S


def rank(L):
    return int(np.linalg.matrix_rank(L))               # exactly the call of model_selection.py:254, :638


def main():
    mg._import_reference()
    from gglasso.helper import model_selection as ms
    from gglasso.helper import data_generation as dg
    from gglasso.solver.single_admm_solver import ADMM_SGL
    from gglasso.solver.admm_solver import ADMM_MGL
    out = {}

    # ---- ADMM_SGL, p = 200
    p, N = 200, 800
    S = latent_cov(dg, p, 5, 1260, N=N)[0]
    mus = np.array([3.0, 1.5, 0.8, 0.4])
    out["sgl_S"], out["sgl_N"], out["sgl_lambda1"], out["sgl_mu1"] = S, np.array(N), np.array(0.03), mus
    ranks, small = [], []
    for mu in mus:
        sol, info = mg.quiet(ADMM_SGL, S, 0.03, np.eye(p), tol=1e-10, rtol=1e-10, latent=True, mu1=float(mu))
        assert info['status'] == 'optimal'
        ranks.append(rank(sol['L']))
        ev = np.linalg.eigvalsh(sol['L'])
        small.append(ev[-ranks[-1]] if ranks[-1] else 0.0)
        if mu == 0.8:
            out["sgl_Theta"], out["sgl_L"] = sol['Theta'], sol['L']
    out["sgl_rank"], out["sgl_smallest_kept"] = np.array(ranks), np.array(small)
    print("ADMM_SGL p=200 ranks", ranks, "smallest kept eigenvalue", np.round(small, 4))

    # ---- single_grid_search on the same S
    lam, mu = np.array([0.05, 0.025]), np.array([1.5, 0.8, 0.4])
    best, est, low, st = mg.quiet(ms.single_grid_search, S, lam, N, method='eBIC', gamma=0.3, latent=True, mu_range=mu,
                                  use_block=False, tol=1e-10, rtol=1e-10)
    gam = sorted(st['BIC'].keys())
    out["sgs_lambda_range"], out["sgs_mu_range"], out["sgs_gammas"] = lam, mu, np.array(gam)
    out["sgs_RANK"], out["sgs_SP"], out["sgs_AIC"] = st['RANK'], st['SP'], st['AIC']
    out["sgs_BIC"] = np.stack([st['BIC'][g] for g in gam])
    out["sgs_best"] = np.array([st['BEST']['lambda1'], st['BEST']['mu1']])
    out["sgs_best_rank"] = np.array(rank(best['L']))
    print("single_grid_search RANK\n", st['RANK'], "best", out["sgs_best"])

    # ---- ADMM_MGL, K = 2, p = 160
    p, K, N = 160, 2, 640
    S2 = latent_cov(dg, p, 5, 1261, K=K, N=N)
    mu1 = np.array([0.8, 0.5])
    out["mgl_S"], out["mgl_N"], out["mgl_mu1"] = S2, np.array(N), mu1
    out["mgl_lambda"] = np.array([0.03, 0.01])
    Om0 = np.stack([np.eye(p)] * K)
    for reg in ("GGL", "FGL"):
        sol, info = mg.quiet(ADMM_MGL, S2, 0.03, 0.01, reg, Om0, tol=1e-10, rtol=1e-10, latent=True, mu1=mu1)
        assert info['status'] == 'optimal'
        rk = [rank(sol['L'][k]) for k in range(K)]
        out[f"mgl_{reg}_rank"], out[f"mgl_{reg}_L"] = np.array(rk), sol['L']
        if reg == "GGL":
            out["mgl_GGL_Theta"] = sol['Theta']
        print(f"ADMM_MGL {reg} K=2 p=160 ranks", rk,
              [np.round(np.linalg.eigvalsh(sol['L'][k])[-rk[k] - 1:-rk[k] + 1 or None], 5) for k in range(K)])

    # ---- grid_search with the reference's ADMM_MGL on that S
    l1, l2 = np.array([0.05, 0.025]), np.array([0.02, 0.008])
    mu_range = np.array([1.5, 0.8, 0.5])
    ix_mu = np.array([[0, 1], [1, 2]])              # (K, len(l1)): mu per instance and lambda1 column (:216-220)
    Nk = np.array([N, N])
    stats, ix, best = mg.quiet(ms.grid_search, ADMM_MGL, S2, Nk, p, "GGL", l1, l2=l2, method='eBIC', gamma=0.3, latent=True,
                               mu_range=mu_range, ix_mu=ix_mu, tol=1e-10, rtol=1e-10)
    out["grid_l1"], out["grid_l2"], out["grid_mu_range"], out["grid_ix_mu"], out["grid_N"] = l1, l2, mu_range, ix_mu, Nk
    out["grid_RANK"], out["grid_SP"], out["grid_ix"] = stats['RANK'], stats['SP'], np.array(ix)
    out["grid_best"] = np.array([stats['BEST']['lambda1'], stats['BEST']['lambda2']])
    out["grid_best_rank"] = np.array([rank(best['L'][k]) for k in range(K)])
    print("grid_search RANK\n", stats['RANK'], "ix", ix)
    mg.save("g18_latent_rank_large_p", **out)


if __name__ == "__main__":
    main()
