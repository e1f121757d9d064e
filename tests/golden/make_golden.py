#!/usr/bin/env python3
"""
Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the real reference (fabian-sp/GGLasso v0.2.1, pure Python) from /root/reference/src with a
throw-away identity-decorator stand-in for ``numba`` written to a temp dir (numba is not installed in
this image; every @njit body in the reference is valid plain NumPy), runs the hot-path functions on
small inputs and stores INPUTS AND OUTPUTS in tests/golden/*.npz.  Nothing of the reference itself
(source, bytecode) is written into the repo; the fixtures are data only.

    python tests/golden/make_golden.py

Sets:
  G1 phiplus, G2 prox_rank_norm, G3 prox_od_1norm, G4 prox_2norm/prox_phi_ggl, G5 condat_method,
  G6 prox_p GGL/FGL, G7 ADMM_stopping_criterion, G8 fixed-length ADMM_MGL trajectories,
  G9 converged ADMM_MGL, G10 ADMM_SGL (+mask, +latent, mask=0 known answer, kkt residuals), G11 block_SGL.
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"


def _import_reference():
    shim = tempfile.mkdtemp(prefix="numba_shim_")
    os.makedirs(os.path.join(shim, "numba"))
    with open(os.path.join(shim, "numba", "__init__.py"), "w") as fh:
        fh.write("def _ident(*a, **k):\n"
                 "    if len(a) == 1 and callable(a[0]) and not k:\n"
                 "        return a[0]\n"
                 "    return lambda f: f\n"
                 "njit = jit = _ident\n")
    with open(os.path.join(shim, "numba", "typed.py"), "w") as fh:
        fh.write("List = list\n")
    sys.path.insert(0, REF_SRC)
    sys.path.insert(0, shim)
    from gglasso.solver import admm_solver, single_admm_solver, ggl_helper, fgl_helper
    from gglasso.helper import data_generation, utils
    return admm_solver, single_admm_solver, ggl_helper, fgl_helper, data_generation, utils


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB  ({len(arrays)} arrays)")


def sym(rng, p, scale=1.0):
    A = rng.standard_normal((p, p)) * scale
    return 0.5 * (A + A.T)


def main():
    admm, sadmm, gh, fh, dg, utils = _import_reference()
    rng = np.random.default_rng(20240917)

    # ------------------------------------------------------------------ G1 / G2
    out = {}
    n = 0
    for p in (4, 16, 32):
        mats = [sym(rng, p), sym(rng, p, 5.0)]
        # repeated eigenvalues: Q diag(1,1,1,2,2,...) Q^T
        Qr, _ = np.linalg.qr(rng.standard_normal((p, p)))
        d = np.repeat(np.arange(1, p // 2 + 2), 2)[:p].astype(float)
        R = (Qr * d) @ Qr.T
        mats.append(0.5 * (R + R.T))
        # near-singular / tiny eigenvalues around zero
        d2 = np.concatenate([np.array([1e-9, -1e-9, 0.0]), rng.standard_normal(p - 3)])
        Ns = (Qr * d2) @ Qr.T
        mats.append(0.5 * (Ns + Ns.T))
        for W in mats:
            for beta in (0.5, 1.0, 2.0):
                D, Q = np.linalg.eigh(W)
                out[f"W_{n}"] = W
                out[f"beta_{n}"] = np.array(beta)
                out[f"phiplus_{n}"] = gh.phiplus(beta, D, Q)
                out[f"rank_{n}"] = gh.prox_rank_norm(W, beta, D, Q)
                n += 1
    out["count"] = np.array(n)
    save("g1_g2_eigen_prox", **out)

    # ------------------------------------------------------------------ G3
    out = {}
    n = 0
    for p in (5, 16, 33):
        A = sym(rng, p)
        mask = np.abs(sym(rng, p))
        mask[rng.random((p, p)) < 0.3] = 0.0
        mask = 0.5 * (mask + mask.T)
        for lam in (0.05, 0.7):
            out[f"A_{n}"] = A
            out[f"lam_{n}"] = np.array(lam)
            out[f"mask_{n}"] = mask
            out[f"scalar_{n}"] = gh.prox_od_1norm(A, lam)
            out[f"masked_{n}"] = gh.prox_od_1norm(A, lam * mask)
            n += 1
    out["count"] = np.array(n)
    save("g3_prox_od_1norm", **out)

    # ------------------------------------------------------------------ G4
    out = {}
    n = 0
    for K in (1, 3, 6):
        for scale in (0.01, 0.3, 2.0):
            v = rng.standard_normal(K) * scale
            for (l1, l2) in ((0.05, 0.1), (0.5, 0.02), (0.1, 5.0)):
                out[f"v_{n}"] = v
                out[f"l_{n}"] = np.array([l1, l2])
                out[f"p2_{n}"] = gh.prox_2norm(v, l2)
                out[f"ggl_{n}"] = gh.prox_phi_ggl(v, l1, l2)
                n += 1
    out["count"] = np.array(n)
    save("g4_group_prox", **out)

    # ------------------------------------------------------------------ G5
    out = {}
    n = 0
    for K in (1, 2, 5, 20, 50):
        signals = [rng.standard_normal(K),
                   np.ones(K) * 0.7,
                   np.linspace(-1, 1, K),
                   np.array([(-1.0) ** i for i in range(K)]),
                   np.round(rng.standard_normal(K), 1),         # ties
                   np.cumsum(rng.standard_normal(K)) * 0.3]
        for y in signals:
            for lam in (1e-3, 0.1, 10.0):
                out[f"y_{n}"] = y
                out[f"lam_{n}"] = np.array(lam)
                out[f"x_{n}"] = fh.condat_method(y.copy(), lam)
                out[f"fgl_{n}"] = gh.prox_phi_fgl(y.copy(), 0.05, lam)
                n += 1
    out["count"] = np.array(n)
    save("g5_condat_tv", **out)

    # ------------------------------------------------------------------ G6
    out = {}
    n = 0
    for (K, p) in ((3, 8), (6, 16), (1, 7)):
        X = rng.standard_normal((K, p, p)) * 0.3
        X = 0.5 * (X + X.transpose(0, 2, 1))
        # lower triangle perturbed within the 1e-5 symmetry assert: prox_p must read the UPPER one
        Xl = X + np.tril(rng.standard_normal((K, p, p)), -1) * 1e-7
        for (l1, l2) in ((0.05, 0.02), (0.2, 0.3)):
            for Xin in (X, Xl):
                out[f"X_{n}"] = Xin
                out[f"l_{n}"] = np.array([l1, l2])
                out[f"ggl_{n}"] = gh.prox_p(Xin, l1, l2, 'GGL')
                out[f"fgl_{n}"] = gh.prox_p(Xin, l1, l2, 'FGL')
                out[f"pval_ggl_{n}"] = np.array(gh.P_val(Xin, l1, l2, 'GGL'))
                out[f"pval_fgl_{n}"] = np.array(gh.P_val(Xin, l1, l2, 'FGL'))
                n += 1
    out["count"] = np.array(n)
    save("g6_prox_p", **out)

    # ------------------------------------------------------------------ problems for G7-G9
    K, p, N = 3, 20, 200
    Sig_g, _ = dg.group_power_network(p, K=K, M=2, seed=1234)
    S_ggl, _ = dg.sample_covariance_matrix(Sig_g, N, seed=1234)
    Sig_f, _ = dg.time_varying_power_network(p, K=K, M=4, seed=1235)
    S_fgl, _ = dg.sample_covariance_matrix(Sig_f, N, seed=1235)
    Om0 = utils.get_K_identity(K, p)
    l1, l2, mu1 = 0.05, 0.01, 0.1

    # ------------------------------------------------------------------ G7
    sol, _ = quiet(admm.ADMM_MGL, S_ggl, l1, l2, 'GGL', Om0, max_iter=3, tol=1e-20, rtol=1e-20)
    sol2, _ = quiet(admm.ADMM_MGL, S_ggl, l1, l2, 'GGL', Om0, max_iter=2, tol=1e-20, rtol=1e-20)
    r = admm.ADMM_stopping_criterion(sol['Omega'], sol2['Omega'], sol['Theta'], sol['L'], sol['X'],
                                     S_ggl, 1.7, 1e-5, 1e-4, False)
    save("g7_stopping", Omega=sol['Omega'], Omega_prev=sol2['Omega'], Theta=sol['Theta'], L=sol['L'],
         X=sol['X'], S=S_ggl, rho=np.array(1.7), eps=np.array([1e-5, 1e-4]), out=np.array(r))

    # ------------------------------------------------------------------ G8 / G9
    out = {"S_GGL": S_ggl, "S_FGL": S_fgl, "Omega_0": Om0,
           "params": np.array([l1, l2, mu1])}
    for reg, S in (("GGL", S_ggl), ("FGL", S_fgl)):
        for latent in (False, True):
            tag = f"{reg}_{'lat' if latent else 'nol'}"
            for mi in (1, 2, 10):
                sol, info = quiet(admm.ADMM_MGL, S, l1, l2, reg, Om0, max_iter=mi, tol=1e-20,
                                  rtol=1e-20, latent=latent, mu1=mu1, measure=True)
                for nm in ('Omega', 'Theta', 'L', 'X'):
                    out[f"{tag}_it{mi}_{nm}"] = sol[nm]
                out[f"{tag}_it{mi}_residual"] = info['residual']
                out[f"{tag}_it{mi}_objective"] = info['objective']
            sol, info = quiet(admm.ADMM_MGL, S, l1, l2, reg, Om0, tol=1e-10, rtol=1e-10,
                              latent=latent, mu1=mu1, measure=True)
            out[f"{tag}_conv_Theta"] = sol['Theta']
            out[f"{tag}_conv_Omega"] = sol['Omega']
            out[f"{tag}_conv_L"] = sol['L']
            out[f"{tag}_conv_iters"] = np.array(len(info['residual']))
            out[f"{tag}_conv_status"] = np.array(info['status'])
            # n_samples weighting (int), no rho update, warm start from a previous solution
            sol2, info2 = quiet(admm.ADMM_MGL, S, l1, l2, reg, sol['Omega'], Theta_0=sol['Theta'],
                                X_0=sol['X'], n_samples=3, max_iter=4, tol=1e-20, rtol=1e-20,
                                update_rho=False, rho=0.7, latent=latent, mu1=mu1)
            for nm in ('Omega', 'Theta', 'L', 'X'):
                out[f"{tag}_warm_{nm}"] = sol2[nm]
                out[f"{tag}_warmstart_{nm}"] = sol[nm]
    # kkt residual of one state (opt-in stopping criterion, admm_solver.py:333)
    sol, _ = quiet(admm.ADMM_MGL, S_ggl, l1, l2, 'GGL', Om0, max_iter=5, tol=1e-20, rtol=1e-20)
    nk = np.ones((K, 1, 1))
    out["kkt_state_Omega"], out["kkt_state_Theta"] = sol['Omega'], sol['Theta']
    out["kkt_state_L"], out["kkt_state_X"] = sol['L'], sol['X']
    out["kkt_value"] = np.array(admm.kkt_stopping_criterion(sol['Omega'], sol['Theta'], sol['L'],
                                                            0.9 * sol['X'], S_ggl, l1, l2, nk, 'GGL'))
    sol, info = quiet(admm.ADMM_MGL, S_ggl, l1, l2, 'GGL', Om0, tol=1e-6, stopping_criterion='kkt',
                      measure=True)
    out["kkt_run_Theta"] = sol['Theta']
    out["kkt_run_iters"] = np.array(len(info['residual']))
    save("g8_g9_admm_mgl", **out)

    # ------------------------------------------------------------------ G10
    p, N = 20, 100
    Sig, _ = dg.generate_precision_matrix(p=p, M=2, style='erdos', prob=0.1, seed=1236)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=1236)
    mask = np.ones((p, p))
    mask[:5, :] = 0.2
    mask[:, :5] = 0.2
    mask[8, 9] = mask[9, 8] = 0.0
    out = {"S": S, "mask": mask, "params": np.array([0.05, 0.2])}
    Om0 = np.eye(p)
    for tag, kw in (("plain", {}), ("mask", {"lambda1_mask": mask}),
                    ("latent", {"latent": True, "mu1": 0.2}),
                    ("zeromask", {"lambda1_mask": np.zeros((p, p))})):
        sol, info = quiet(sadmm.ADMM_SGL, S, 0.05, Om0, max_iter=10, tol=1e-20, rtol=1e-20,
                          measure=True, **kw)
        for nm in sol:
            out[f"{tag}_it10_{nm}"] = sol[nm]
        out[f"{tag}_it10_residual"] = info['residual']
        sol, info = quiet(sadmm.ADMM_SGL, S, 0.05, Om0, tol=1e-10, rtol=1e-10, measure=True, **kw)
        out[f"{tag}_conv_Theta"] = sol['Theta']
        out[f"{tag}_conv_iters"] = np.array(len(info['residual']))
        out[f"{tag}_conv_status"] = np.array(info['status'])
    out["inv_S"] = np.linalg.inv(S)     # known answer for zeromask (reference tests/test_solvers.py:191-216)
    sol, _ = quiet(sadmm.ADMM_SGL, S, 0.05, Om0, max_iter=5, tol=1e-20, rtol=1e-20)
    out["kkt_state_Omega"], out["kkt_state_Theta"], out["kkt_state_X"] = sol['Omega'], sol['Theta'], sol['X']
    out["kkt_value"] = np.array(sadmm.kkt_stopping_criterion(sol['Omega'], sol['Theta'], np.zeros((p, p)),
                                                             0.8 * sol['X'], S, 0.05))
    save("g10_admm_sgl", **out)

    # ------------------------------------------------------------------ G11 block_SGL
    # block-diagonal truth + large lambda1 => several connected components incl. singletons
    p, N = 30, 200
    Sig, _ = dg.generate_precision_matrix(p=p, M=5, style='erdos', prob=0.3, seed=1237)
    S, _ = dg.sample_covariance_matrix(Sig, N, seed=1237)
    lam = 0.25
    numC, allC = sadmm.get_connected_components(S, lam)
    sol = quiet(sadmm.block_SGL, S, lam, np.eye(p), tol=1e-10, rtol=1e-10)
    full, _ = quiet(sadmm.ADMM_SGL, S, lam, np.eye(p), tol=1e-10, rtol=1e-10)
    mask = np.ones((p, p))
    mask[:6, :] = mask[:, :6] = 0.5
    solm = quiet(sadmm.block_SGL, S, lam, np.eye(p), tol=1e-10, rtol=1e-10, lambda1_mask=mask)
    save("g11_block_sgl", S=S, lam=np.array(lam), numC=np.array(numC),
         sizes=np.array(sorted(len(c) for c in allC)), Omega=sol['Omega'], Theta=sol['Theta'], X=sol['X'],
         full_Theta=full['Theta'], mask=mask, mask_Theta=solm['Theta'], mask_Omega=solm['Omega'])


if __name__ == "__main__":
    main()
