#!/usr/bin/env python3
"""
Golden vectors for ext_ADMM_MGL, generated like make_golden.py by importing the REAL
reference in the build container (solver/ext_admm_solver.py:18-453, helper/ext_admm_helper.py:46-144):

  G14 a NON-CONFORMING problem: K = 4 instances over a universe of 13 variables, every instance holding its own subset
      (p_k = 8, 10, 9, 11); the bookkeeping array G comes from the reference's own construct_indexer /
      create_group_array.  Stored: the S_k, G, and the reference's Omega, Theta, L, X0, X1 after 1, 2 and 10 iterations
      (tol = rtol = 1e-20, so the length is fixed), the residual histories, the converged solution at
      tol = rtol = 1e-9 with its status and iteration count -- with and without latent variables -- a warm start
      with X0 / X1 given and per-instance lambda1 / mu1, and a run under the KKT criterion (prox_2norm_G operator
      vectors included).
  G15 the CONFORMING case of the reference's own consistency test (tests/test_solvers.py:71-120, scaled down):
      trivial G, ext_ADMM_MGL(lambda2 / sqrt(K)) next to ADMM_MGL(lambda2).

    python tests/golden/make_golden_ext.py
"""
import numpy as np

import make_golden as mg


def _stack_out(out, tag, sol, p):
    for nm in ("Omega", "Theta", "L", "X0", "X1"):
        for k in range(len(p)):
            out[f"{tag}_{nm}_{k}"] = sol[nm][k]


def main():
    mg._import_reference()
    import pandas as pd
    from gglasso.solver import ext_admm_solver as ext
    from gglasso.solver import admm_solver as admm
    from gglasso.helper import ext_admm_helper as eh
    from gglasso.helper import data_generation as dg

    rng = np.random.default_rng(20241002)
    # ---------------------------------------------------------------- G14: non-conforming instances
    universe = np.arange(13)
    members = [np.sort(rng.choice(universe, size=n, replace=False)) for n in (8, 10, 9, 11)]
    K = len(members)
    # one sparse precision over the universe, every instance observes its own variables (different sample sizes)
    Sig, _ = dg.generate_precision_matrix(p=13, M=1, style='erdos', prob=0.3, seed=77)
    samples, S = [], {}
    for k, ix in enumerate(members):
        N = 60 + 20 * k
        X = rng.multivariate_normal(np.zeros(13), Sig, size=N).T[ix]
        samples.append(pd.DataFrame(X, index=ix))
        S[k] = np.cov(X, bias=True)
    ix_exist, ix_location = eh.construct_indexer(samples)
    G = mg.quiet(eh.create_group_array, ix_exist, ix_location, 2)
    p = np.array([len(ix) for ix in members])
    eh.check_G(G, p)
    out = {"K": np.array(K), "p": p, "G": G, "params": np.array([0.08, 0.05, 0.4])}
    for k in range(K):
        out[f"S_{k}"] = S[k]
        out[f"members_{k}"] = members[k]
    l1, l2, mu1 = 0.08, 0.05, 0.4
    Om0 = eh.get_K_identity(p)

    # the group shrink on its own (operator vectors): symmetric inputs, two thresholds
    Z = {k: mg.sym(rng, p[k]) for k in range(K)}
    for n, lam in enumerate((0.05, 0.6)):
        res = ext.prox_2norm_G({k: Z[k].copy() for k in range(K)}, G, lam)
        for k in range(K):
            out[f"proxG_in_{k}"] = Z[k]
            out[f"proxG{n}_out_{k}"] = res[k]
        out[f"proxG{n}_lam"] = np.array(lam)

    for latent in (False, True):
        tag = "lat" if latent else "nol"
        kw = dict(latent=latent, mu1=mu1)
        for mi in (1, 2, 10):
            sol, info = mg.quiet(ext.ext_ADMM_MGL, S, l1, l2, 'GGL', {k: v.copy() for k, v in Om0.items()}, G,
                                 max_iter=mi, tol=1e-20, rtol=1e-20, measure=True, **kw)
            _stack_out(out, f"{tag}_it{mi}", sol, p)
            out[f"{tag}_it{mi}_residual"] = info['residual']
            assert info['status'] == 'max iterations reached'
        sol, info = mg.quiet(ext.ext_ADMM_MGL, S, l1, l2, 'GGL', {k: v.copy() for k, v in Om0.items()}, G, tol=1e-9,
                             rtol=1e-9, measure=True, **kw)
        _stack_out(out, f"{tag}_conv", sol, p)
        out[f"{tag}_conv_status"] = np.array(info['status'])
        out[f"{tag}_conv_iters"] = np.array(len(info['residual']))
        # warm start with both duals given, per-instance lambda1 / mu1, another rho
        lam1_k = np.array([0.05, 0.08, 0.11, 0.07])
        mu_k = np.array([0.3, 0.5, 0.4, 0.6])
        X0w = {k: 0.3 * sol['X0'][k] for k in range(K)}
        X1w = {k: 0.5 * sol['X1'][k] for k in range(K)}
        Omw = {k: sol['Omega'][k].copy() for k in range(K)}
        solw, infow = mg.quiet(ext.ext_ADMM_MGL, S, lam1_k, l2, 'GGL', Omw, G, X0={k: v.copy() for k, v in X0w.items()},
                               X1={k: v.copy() for k, v in X1w.items()}, rho=1.7, max_iter=5, tol=1e-20, rtol=1e-20,
                               measure=True, latent=latent, mu1=mu_k)
        _stack_out(out, f"{tag}_warm", solw, p)
        out[f"{tag}_warm_residual"] = infow['residual']
        for k in range(K):
            out[f"{tag}_warmstart_Omega_{k}"], out[f"{tag}_warmstart_X0_{k}"] = sol['Omega'][k], X0w[k]
            out[f"{tag}_warmstart_X1_{k}"] = X1w[k]
        out["warm_lambda1"], out["warm_mu1"] = lam1_k, mu_k
        # KKT criterion
        solk, infok = mg.quiet(ext.ext_ADMM_MGL, S, l1, l2, 'GGL', {k: v.copy() for k, v in Om0.items()}, G,
                               stopping_criterion='kkt', tol=1e-6, measure=True, **kw)
        _stack_out(out, f"{tag}_kkt", solk, p)
        out[f"{tag}_kkt_residual"] = infok['residual']
        out[f"{tag}_kkt_status"] = np.array(infok['status'])
    mg.save("g14_ext_admm_nonconforming", **out)

    # ---------------------------------------------------------------- G15: conforming case (trivial G)
    p2, K2, N2 = 20, 3, 400
    Sigma, _ = dg.group_power_network(p2, K2, 4, seed=1234)
    S2, _ = dg.sample_covariance_matrix(Sigma, N2, seed=1234)
    Sd = {k: S2[k].copy() for k in range(K2)}
    G2 = eh.construct_trivial_G(p2, K2)
    out = {"S": S2, "G": G2, "params": np.array([0.05, 0.01, 0.1])}
    for latent in (False, True):
        tag = "lat" if latent else "nol"
        sol, info = mg.quiet(ext.ext_ADMM_MGL, Sd, 0.05, 0.01 / np.sqrt(K2), 'GGL', eh.get_K_identity(np.full(K2, p2)), G2,
                             tol=1e-9, rtol=1e-9, latent=latent, mu1=0.1, measure=True)
        for nm in ("Omega", "Theta", "L", "X0", "X1"):
            out[f"{tag}_ext_{nm}"] = np.stack([sol[nm][k] for k in range(K2)])
        out[f"{tag}_ext_iters"] = np.array(len(info['residual']))
        out[f"{tag}_ext_status"] = np.array(info['status'])
        solm, _ = mg.quiet(admm.ADMM_MGL, S2, 0.05, 0.01, 'GGL', np.stack([np.eye(p2)] * K2), tol=1e-9, rtol=1e-9,
                           latent=latent, mu1=0.1)
        out[f"{tag}_mgl_Theta"], out[f"{tag}_mgl_L"] = solm['Theta'], solm['L']
    mg.save("g15_ext_admm_conforming", **out)


if __name__ == "__main__":
    main()
