"""Pins the CPU oracle (oracle/ggl_oracle.py + oracle/ggl_oracle.c) to the golden vectors captured
from the imported reference by tests/golden/make_golden.py.  CPU only."""
import subprocess
import os

import numpy as np
import pytest

from conftest import load_golden, ROOT
from oracle import ggl_oracle as orc


@pytest.fixture(scope="module", autouse=True)
def _build_c_oracle():
    subprocess.check_call([os.path.join(ROOT, "oracle", "build.sh")])
    orc._CLIB = None
    assert orc._clib() is not None


def test_g1_g2_phiplus_rank():
    g = load_golden("g1_g2_eigen_prox")
    for n in range(int(g["count"])):
        W, beta = g[f"W_{n}"], float(g[f"beta_{n}"])
        D, Q = np.linalg.eigh(W)
        scale = max(1.0, np.abs(W).max())
        assert np.abs(orc.phiplus(beta, D, Q) - g[f"phiplus_{n}"]).max() <= 1e-12 * scale
        assert np.abs(orc.prox_rank_norm(W, beta, D, Q) - g[f"rank_{n}"]).max() <= 1e-12 * scale
        om, _ = orc.phiplus_stack(W[None], beta)
        assert np.abs(om[0] - g[f"phiplus_{n}"]).max() <= 1e-12 * scale
        assert np.abs(orc.rank_stack(W[None], beta)[0] - g[f"rank_{n}"]).max() <= 1e-12 * scale


def test_g3_prox_od_1norm():
    g = load_golden("g3_prox_od_1norm")
    for n in range(int(g["count"])):
        A, lam, mask = g[f"A_{n}"], float(g[f"lam_{n}"]), g[f"mask_{n}"]
        assert np.array_equal(orc.prox_od_1norm(A, lam), g[f"scalar_{n}"])
        assert np.array_equal(orc.prox_od_1norm(A, lam * mask), g[f"masked_{n}"])


def test_g4_group_prox():
    g = load_golden("g4_group_prox")
    for n in range(int(g["count"])):
        v, (l1, l2) = g[f"v_{n}"], g[f"l_{n}"]
        assert np.abs(orc.prox_2norm(v, l2) - g[f"p2_{n}"]).max() <= 1e-15
        assert np.abs(orc.prox_phi_ggl(v, l1, l2) - g[f"ggl_{n}"]).max() <= 1e-15


def test_g5_condat_python_and_c():
    import ctypes
    g = load_golden("g5_condat_tv")
    lib = orc._clib()
    dp = ctypes.POINTER(ctypes.c_double)
    for n in range(int(g["count"])):
        y, lam = g[f"y_{n}"], float(g[f"lam_{n}"])
        assert np.array_equal(orc.condat_method(y, lam), g[f"x_{n}"])
        assert np.array_equal(orc.prox_phi_fgl(y, 0.05, lam), g[f"fgl_{n}"])
        yc = np.ascontiguousarray(y)
        x = np.empty_like(yc)
        lib.oracle_condat(yc.ctypes.data_as(dp), x.ctypes.data_as(dp), len(yc), lam)
        assert np.array_equal(x, g[f"x_{n}"])


def test_g6_prox_p_all_variants():
    g = load_golden("g6_prox_p")
    for n in range(int(g["count"])):
        X, (l1, l2) = g[f"X_{n}"], g[f"l_{n}"]
        for reg, key in (("GGL", "ggl"), ("FGL", "fgl")):
            ref = g[f"{key}_{n}"]
            assert np.abs(orc.prox_p(X, l1, l2, reg) - ref).max() <= 1e-14
            assert np.abs(orc.prox_p_loops(X, l1, l2, reg) - ref).max() <= 1e-14
            assert np.abs(orc.prox_p_c(X, l1, l2, reg) - ref).max() <= 1e-14
            assert abs(orc.P_val(X, l1, l2, reg) - float(g[f"pval_{key}_{n}"])) <= 1e-12


def test_g7_stopping_criterion():
    g = load_golden("g7_stopping")
    r = orc.ADMM_stopping_criterion(g["Omega"], g["Omega_prev"], g["Theta"], g["L"], g["X"], g["S"],
                                    float(g["rho"]), *g["eps"], False)
    assert np.allclose(r, g["out"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("reg", ["GGL", "FGL"])
@pytest.mark.parametrize("latent", [False, True])
def test_g8_g9_admm_mgl(reg, latent):
    g = load_golden("g8_g9_admm_mgl")
    S, Om0 = g[f"S_{reg}"], g["Omega_0"]
    l1, l2, mu1 = g["params"]
    tag = f"{reg}_{'lat' if latent else 'nol'}"
    for mi in (1, 2, 10):
        sol, info = orc.ADMM_MGL(S, l1, l2, reg, Om0, max_iter=mi, tol=1e-20, rtol=1e-20,
                                 latent=latent, mu1=float(mu1), measure=True)
        for nm in ('Omega', 'Theta', 'L', 'X'):
            assert np.abs(sol[nm] - g[f"{tag}_it{mi}_{nm}"]).max() <= 1e-10, (mi, nm)
        assert np.allclose(info['residual'], g[f"{tag}_it{mi}_residual"], rtol=1e-8)
        assert np.allclose(info['objective'], g[f"{tag}_it{mi}_objective"], rtol=1e-10)
        assert info['status'] == 'max iterations reached'
    sol, info = orc.ADMM_MGL(S, l1, l2, reg, Om0, tol=1e-10, rtol=1e-10, latent=latent, mu1=float(mu1))
    assert np.linalg.norm(sol['Theta'] - g[f"{tag}_conv_Theta"]) <= 1e-8
    assert info['status'] == str(g[f"{tag}_conv_status"])
    assert abs(info['iterations'] - int(g[f"{tag}_conv_iters"])) <= 1
    # n_samples weighting, fixed rho, warm start
    sol2, _ = orc.ADMM_MGL(S, l1, l2, reg, g[f"{tag}_warmstart_Omega"], Theta_0=g[f"{tag}_warmstart_Theta"],
                           X_0=g[f"{tag}_warmstart_X"], n_samples=3, max_iter=4, tol=1e-20, rtol=1e-20,
                           update_rho=False, rho=0.7, latent=latent, mu1=float(mu1))
    for nm in ('Omega', 'Theta', 'L', 'X'):
        assert np.abs(sol2[nm] - g[f"{tag}_warm_{nm}"]).max() <= 1e-10


def test_g9_kkt():
    g = load_golden("g8_g9_admm_mgl")
    S = g["S_GGL"]
    l1, l2, _ = g["params"]
    nk = np.ones((S.shape[0], 1, 1))
    v = orc.kkt_stopping_criterion_mgl(g["kkt_state_Omega"], g["kkt_state_Theta"], g["kkt_state_L"],
                                       0.9 * g["kkt_state_X"], S, l1, l2, nk, 'GGL')
    assert abs(v - float(g["kkt_value"])) <= 1e-12
    sol, info = orc.ADMM_MGL(S, l1, l2, 'GGL', g["Omega_0"], tol=1e-6, stopping_criterion='kkt')
    assert info['iterations'] == int(g["kkt_run_iters"])
    assert np.abs(sol['Theta'] - g["kkt_run_Theta"]).max() <= 1e-10


@pytest.mark.parametrize("tag", ["plain", "mask", "latent", "zeromask"])
def test_g10_admm_sgl(tag):
    g = load_golden("g10_admm_sgl")
    S, mask = g["S"], g["mask"]
    p = S.shape[0]
    kw = {"plain": {}, "mask": {"lambda1_mask": mask}, "latent": {"latent": True, "mu1": 0.2},
          "zeromask": {"lambda1_mask": np.zeros((p, p))}}[tag]
    sol, info = orc.ADMM_SGL(S, 0.05, np.eye(p), max_iter=10, tol=1e-20, rtol=1e-20, measure=True, **kw)
    for nm in sol:
        assert np.abs(sol[nm] - g[f"{tag}_it10_{nm}"]).max() <= 1e-10
    assert ('L' in sol) == (tag == "latent")
    assert np.allclose(info['residual'], g[f"{tag}_it10_residual"], rtol=1e-8)
    sol, info = orc.ADMM_SGL(S, 0.05, np.eye(p), tol=1e-10, rtol=1e-10, **kw)
    assert np.linalg.norm(sol['Theta'] - g[f"{tag}_conv_Theta"]) <= 1e-8
    assert info['status'] == str(g[f"{tag}_conv_status"])
    if tag == "zeromask":
        # reference known-answer test (tests/test_solvers.py:191-216): no penalty => Theta = inv(S)
        assert np.abs(sol['Theta'] - g["inv_S"]).max() <= 1e-4


def test_g10_kkt_sgl():
    g = load_golden("g10_admm_sgl")
    p = g["S"].shape[0]
    v = orc.kkt_stopping_criterion_sgl(g["kkt_state_Omega"], g["kkt_state_Theta"], np.zeros((p, p)),
                                       0.8 * g["kkt_state_X"], g["S"], 0.05)
    assert abs(v - float(g["kkt_value"])) <= 1e-12


def test_g11_block_sgl():
    g = load_golden("g11_block_sgl")
    S, lam = g["S"], float(g["lam"])
    p = S.shape[0]
    numC, allC = orc.get_connected_components(S, lam)
    assert numC == int(g["numC"]) and numC > 1
    assert sorted(len(c) for c in allC) == list(g["sizes"])
    sol = orc.block_SGL(S, lam, np.eye(p), tol=1e-10, rtol=1e-10)
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(sol[nm] - g[nm]).max() <= 1e-9, nm
    # reference test tests/test_solvers.py:123-148: block solution == full solution (3 decimals)
    assert np.abs(sol["Theta"] - g["full_Theta"]).max() <= 1e-3
    solm = orc.block_SGL(S, lam, np.eye(p), tol=1e-10, rtol=1e-10, lambda1_mask=g["mask"])
    assert np.abs(solm["Theta"] - g["mask_Theta"]).max() <= 1e-9


# ---- ext_ADMM_MGL (instances of different dimension; fixtures G14 / G15 from the real reference) ---------------

def test_g14_prox_2norm_G():
    g = load_golden("g14_ext_admm_nonconforming")
    K, G = int(g["K"]), g["G"]
    orc.check_G(G, g["p"])
    for n in range(2):
        res = orc.prox_2norm_G({k: g[f"proxG_in_{k}"].copy() for k in range(K)}, G, float(g[f"proxG{n}_lam"]))
        vec = orc.prox_2norm_G_vectorized({k: g[f"proxG_in_{k}"].copy() for k in range(K)}, G, float(g[f"proxG{n}_lam"]))
        for k in range(K):
            assert np.abs(res[k] - g[f"proxG{n}_out_{k}"]).max() <= 1e-15
            assert np.array_equal(res[k], res[k].T)
            assert np.abs(vec[k] - g[f"proxG{n}_out_{k}"]).max() <= 1e-15       # all groups at once == the loop
    assert orc._G_entries_distinct(G, g["p"])


@pytest.mark.parametrize("latent", [False, True])
def test_g14_ext_admm_oracle(latent):
    import ext_checks
    ext_checks.check_g14(load_golden, orc.ext_ADMM_MGL_printing, latent)
    ext_checks.check_g14_kkt(load_golden, orc.ext_ADMM_MGL_printing, latent)


@pytest.mark.parametrize("latent", [False, True])
def test_g15_ext_admm_conforming_oracle(latent):
    import ext_checks
    ext_checks.check_g15(load_golden, orc.ext_ADMM_MGL_printing, latent)
    g = load_golden("g15_ext_admm_conforming")
    assert np.array_equal(orc.construct_trivial_G(g["S"].shape[1], g["S"].shape[0]), g["G"])


def test_g18_latent_rank_above_the_jacobi_limit():
    """G18 (tests/golden/make_golden_rank.py): the reference's numpy.linalg.matrix_rank(sol['L']) at p = 200 / 160
    (helper/model_selection.py:254, :638) and the entries of Theta / L -- the oracle the GPU tests of the same fixture use."""
    g = load_golden("g18_latent_rank_large_p")
    S, lam = g["sgl_S"], float(g["sgl_lambda1"])
    p = S.shape[0]
    for mu, want in zip(g["sgl_mu1"], g["sgl_rank"]):
        sol, info = orc.ADMM_SGL(S, lam, np.eye(p), tol=1e-10, rtol=1e-10, latent=True, mu1=float(mu))
        assert info['status'] == 'optimal'
        assert np.linalg.matrix_rank(sol['L']) == want
        if mu == 0.8:
            assert np.linalg.norm(sol['Theta'] - g["sgl_Theta"]) <= 1e-8
            assert np.linalg.norm(sol['L'] - g["sgl_L"]) <= 1e-8
    S2, (l1, l2), mu1 = g["mgl_S"], g["mgl_lambda"], g["mgl_mu1"]
    K, p = S2.shape[:2]
    for reg in ("GGL", "FGL"):
        sol, info = orc.ADMM_MGL(S2, l1, l2, reg, np.stack([np.eye(p)] * K), tol=1e-10, rtol=1e-10, latent=True, mu1=mu1)
        assert [np.linalg.matrix_rank(sol['L'][k]) for k in range(K)] == list(g[f"mgl_{reg}_rank"])
        assert np.linalg.norm(sol['L'] - g[f"mgl_{reg}_L"]) <= 1e-8
        if reg == "GGL":
            assert np.linalg.norm(sol['Theta'] - g["mgl_GGL_Theta"]) <= 1e-8
