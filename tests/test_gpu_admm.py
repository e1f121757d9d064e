"""ADMM-level parity on the MI355X: fixed-length trajectories, converged solutions, warm starts,
status strings and the reference's known-answer test, all through the C ABI."""
import contextlib
import io

import numpy as np
import pytest

from conftest import load_golden
from oracle import ggl_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sol():
    from gglasso_amd import solver
    return solver


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


@pytest.mark.parametrize("reg", ["GGL", "FGL"])
@pytest.mark.parametrize("latent", [False, True])
def test_g8_fixed_length_trajectories(sol, reg, latent):
    g = load_golden("g8_g9_admm_mgl")
    S, Om0 = g[f"S_{reg}"], g["Omega_0"]
    l1, l2, mu1 = g["params"]
    tag = f"{reg}_{'lat' if latent else 'nol'}"
    for mi in (1, 2, 10):
        (s, info), out = quiet(sol.ADMM_MGL, S, l1, l2, reg, Om0, max_iter=mi, tol=1e-20, rtol=1e-20,
                               latent=latent, mu1=float(mu1), measure=True)
        for nm in ('Omega', 'Theta', 'L', 'X'):
            assert np.abs(s[nm] - g[f"{tag}_it{mi}_{nm}"]).max() <= 1e-10, (mi, nm)
        assert np.allclose(info['residual'], g[f"{tag}_it{mi}_residual"], rtol=1e-8)
        assert np.allclose(info['objective'], g[f"{tag}_it{mi}_objective"], rtol=1e-10)
        assert info['status'] == 'max iterations reached'
        assert f"ADMM terminated after {mi} iterations with status: max iterations reached." in out
        assert len(info['runtime']) == mi


@pytest.mark.parametrize("reg", ["GGL", "FGL"])
@pytest.mark.parametrize("latent", [False, True])
def test_g9_converged_and_warm_start(sol, reg, latent):
    g = load_golden("g8_g9_admm_mgl")
    S, Om0 = g[f"S_{reg}"], g["Omega_0"]
    l1, l2, mu1 = g["params"]
    tag = f"{reg}_{'lat' if latent else 'nol'}"
    (s, info), _ = quiet(sol.ADMM_MGL, S, l1, l2, reg, Om0, tol=1e-10, rtol=1e-10, latent=latent,
                         mu1=float(mu1), measure=True)
    assert np.linalg.norm(s['Theta'] - g[f"{tag}_conv_Theta"]) <= 1e-8      # the north-star tolerance
    assert info['status'] == str(g[f"{tag}_conv_status"])
    assert abs(len(info['residual']) - int(g[f"{tag}_conv_iters"])) <= 1
    (s2, _), _ = quiet(sol.ADMM_MGL, S, l1, l2, reg, g[f"{tag}_warmstart_Omega"],
                       Theta_0=g[f"{tag}_warmstart_Theta"], X_0=g[f"{tag}_warmstart_X"], n_samples=3,
                       max_iter=4, tol=1e-20, rtol=1e-20, update_rho=False, rho=0.7, latent=latent,
                       mu1=float(mu1))
    for nm in ('Omega', 'Theta', 'L', 'X'):
        assert np.abs(s2[nm] - g[f"{tag}_warm_{nm}"]).max() <= 1e-10


def test_kkt_stopping(sol):
    g = load_golden("g8_g9_admm_mgl")
    S = g["S_GGL"]
    l1, l2, _ = g["params"]
    (s, info), _ = quiet(sol.ADMM_MGL, S, l1, l2, 'GGL', g["Omega_0"], tol=1e-6, stopping_criterion='kkt',
                         measure=True)
    assert len(info['residual']) == int(g["kkt_run_iters"])
    assert np.abs(s['Theta'] - g["kkt_run_Theta"]).max() <= 1e-10


@pytest.mark.parametrize("tag", ["plain", "mask", "latent", "zeromask"])
def test_g10_admm_sgl(sol, tag):
    g = load_golden("g10_admm_sgl")
    S, mask = g["S"], g["mask"]
    p = S.shape[0]
    kw = {"plain": {}, "mask": {"lambda1_mask": mask}, "latent": {"latent": True, "mu1": 0.2},
          "zeromask": {"lambda1_mask": np.zeros((p, p))}}[tag]
    (s, info), _ = quiet(sol.ADMM_SGL, S, 0.05, np.eye(p), max_iter=10, tol=1e-20, rtol=1e-20, measure=True, **kw)
    assert ('L' in s) == (tag == "latent")
    for nm in s:
        assert np.abs(s[nm] - g[f"{tag}_it10_{nm}"]).max() <= 1e-10, nm
    assert np.allclose(info['residual'], g[f"{tag}_it10_residual"], rtol=1e-8)
    assert 'objective' not in info
    (s, info), _ = quiet(sol.ADMM_SGL, S, 0.05, np.eye(p), tol=1e-10, rtol=1e-10, **kw)
    assert np.linalg.norm(s['Theta'] - g[f"{tag}_conv_Theta"]) <= 1e-8
    assert info['status'] == str(g[f"{tag}_conv_status"])
    if tag == "zeromask":
        assert np.abs(s['Theta'] - g["inv_S"]).max() <= 1e-4     # reference tests/test_solvers.py:191-216


def test_sgl_kkt(sol):
    g = load_golden("g10_admm_sgl")
    S = g["S"]
    p = S.shape[0]
    (s, info), _ = quiet(sol.ADMM_SGL, S, 0.05, np.eye(p), tol=1e-7, stopping_criterion='kkt', measure=True)
    ref, rinfo = orc.ADMM_SGL(S, 0.05, np.eye(p), tol=1e-7, stopping_criterion='kkt', measure=True)
    assert len(info['residual']) == rinfo['iterations']
    assert np.abs(s['Theta'] - ref['Theta']).max() <= 1e-10


def test_status_strings_like_reference_tests(sol):
    """reference tests/test_solvers.py:25-65: p=50, K=3 -> 'optimal' at tol=rtol=1e-5, two iterations ->
    'max iterations reached'."""
    from gglasso_amd import synth
    for reg in ("GGL", "FGL"):
        S, _ = synth.make_problem(reg, K=3, p=50, N=1000, seed=3)
        Om0 = np.stack([np.eye(50)] * 3)
        for latent in (False, True):
            (s, info), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, reg, Om0, tol=1e-5, rtol=1e-5, latent=latent, mu1=0.01)
            assert info['status'] == 'optimal'
            (s, info), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, reg, Om0, max_iter=2, latent=latent, mu1=0.01)
            assert info['status'] == 'max iterations reached'


@pytest.mark.parametrize("reg,K,p,latent", [("GGL", 4, 96, False), ("FGL", 5, 90, True), ("GGL", 3, 160, False),
                                            ("FGL", 40, 40, False), ("GGL", 6, 150, True),
                                            # odd p: the direct-to-LDS product kernel needs 16-byte rows, k_symm_tn steps in
                                            ("GGL", 3, 151, False), ("FGL", 7, 133, True)])
def test_oracle_trajectory_mid_sizes(sol, reg, K, p, latent):
    """Seeded problems larger than the fixtures, HIP path vs CPU oracle (covers the LDS-Jacobi /
    rocSOLVER+MFMA switch at p = 128 and the FGL tile switch at K = 32)."""
    from gglasso_amd import synth
    S, _ = synth.make_problem(reg, K=K, p=p, N=2 * p, seed=17)
    Om0 = np.stack([np.eye(p)] * K)
    ref, rinfo = orc.ADMM_MGL(S, 0.05, 0.01, reg, Om0, max_iter=12, tol=1e-20, rtol=1e-20, latent=latent, mu1=0.1)
    (s, info), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, reg, Om0, max_iter=12, tol=1e-20, rtol=1e-20, latent=latent,
                         mu1=0.1)
    for nm in ('Omega', 'Theta', 'L', 'X'):
        assert np.abs(s[nm] - ref[nm]).max() <= 1e-9, nm
    ref, rinfo = orc.ADMM_MGL(S, 0.05, 0.01, reg, Om0, tol=1e-9, rtol=1e-9, latent=latent, mu1=0.1)
    (s, info), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, reg, Om0, tol=1e-9, rtol=1e-9, latent=latent, mu1=0.1)
    assert info['status'] == rinfo['status']
    assert np.linalg.norm(s['Theta'] - ref['Theta']) <= 1e-8


@pytest.mark.parametrize("p,K", [(60, 4), (160, 4), (70, 12), (150, 20), (66, 32),
                                 # K > 32: the K-column of an element over 8 / 16 waves (theta_pair.hip, launch_flat4_any):
                                 # 8 x 8 (K <= 64), 16 x 8 (<= 128), 16 x 16 with 8-byte accesses (<= 256); even and odd p
                                 (40, 40), (33, 64), (36, 100), (31, 128), (30, 200), (25, 256)])
def test_asymmetric_dual_start_takes_the_mirroring_kernels(sol, p, K, monkeypatch):
    """The per-element GGL Theta-step is only valid for a bitwise symmetric state.  A dual start X_0 that is
    symmetric only to ~1e-8 (inside the reference's own 1e-5 assert, ggl_helper.py:193) must follow the
    reference's 'upper triangle, then mirror' semantics: the ctx detects the asymmetry when the state is set and
    runs the tile-pair kernels.  Both dispatches agree with the oracle; for a symmetric start they agree with
    each other to rounding."""
    from gglasso_amd import synth
    S, _ = synth.make_problem("GGL", K=K, p=p, N=2 * p, seed=23)
    Om0 = np.stack([np.eye(p)] * K)
    rng = np.random.default_rng(5)
    X0 = 0.01 * rng.standard_normal((K, p, p))
    X0 = 0.5 * (X0 + X0.transpose(0, 2, 1))
    Xa = X0 + 1e-8 * np.triu(rng.standard_normal((K, p, p)), 1)           # upper triangle perturbed
    for X_start in (X0, Xa):
        ref, _ = orc.ADMM_MGL(S, 0.05, 0.01, "GGL", Om0, X_0=X_start, max_iter=10, tol=1e-20, rtol=1e-20)
        (s, _), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, "GGL", Om0, X_0=X_start, max_iter=10, tol=1e-20, rtol=1e-20)
        for nm in ('Omega', 'Theta', 'X'):
            assert np.abs(s[nm] - ref[nm]).max() <= 1e-9, nm
        assert np.array_equal(s['Theta'], s['Theta'].transpose(0, 2, 1))
    outs = []
    for flat in (0, 1, 2):      # tile-pair kernels / per-element kernel / per-element with the K-column over four waves
        monkeypatch.setitem(sol.ENGINE_OPTIONS, "theta_flat", flat)
        (sx, _), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, "GGL", Om0, X_0=X0, max_iter=10, tol=1e-20, rtol=1e-20)
        outs.append(sx)
    assert np.abs(outs[0]['Theta'] - outs[1]['Theta']).max() <= 1e-12
    assert np.abs(outs[0]['Theta'] - outs[2]['Theta']).max() <= 1e-12


def test_inputs_not_mutated_and_asserts(sol):
    g = load_golden("g8_g9_admm_mgl")
    S, Om0 = g["S_GGL"].copy(), g["Omega_0"].copy()
    S0, O0 = S.copy(), Om0.copy()
    quiet(sol.ADMM_MGL, S, 0.05, 0.01, 'GGL', Om0, max_iter=3)
    assert np.array_equal(S, S0) and np.array_equal(Om0, O0)
    with pytest.raises(AssertionError):
        sol.ADMM_MGL(S, 0.05, 0.01, 'XYZ', Om0)
    with pytest.raises(AssertionError):
        sol.ADMM_MGL(S, 0.05, -1.0, 'GGL', Om0)
    with pytest.raises(AssertionError):
        sol.ADMM_MGL(S, 0.05, 0.01, 'GGL', Om0, rho=0.0)
    with pytest.raises(AssertionError):
        sol.ADMM_MGL(S, 0.05, 0.01, 'GGL', Om0, latent=True)          # mu1 missing


# ---- batched lambda path (K independent ADMM_SGL problems in one ctx) ----------------------------------

@pytest.mark.parametrize("p,latent", [(30, False), (30, True), (150, False), (140, True)])
def test_sgl_batch_equals_independent_solves(p, latent):
    """Every instance of the batch must follow the trajectory of its own ADMM_SGL call (own rho, own
    stopping iteration) -- checked against the CPU oracle at p below and above the LDS-Jacobi limit."""
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_SGL_batch
    S, _ = synth.make_problem("GGL", 1, p, N=3 * p, seed=5)
    S = S[0]
    lams = np.logspace(0, -2, 6)
    mu1 = 0.3 if latent else None
    res = ADMM_SGL_batch(S, lams, tol=1e-9, rtol=1e-9, latent=latent, mu1=mu1)
    assert len(res) == len(lams)
    for k, lam in enumerate(lams):
        ref, rinfo = orc.ADMM_SGL(S, lam, np.eye(p), tol=1e-9, rtol=1e-9, latent=latent, mu1=mu1)
        sol, info = res[k]
        assert info['status'] == rinfo['status'] == 'optimal'
        assert info['iterations'] == rinfo['iterations'], (k, lam)
        assert abs(info['rho'] - rinfo['rho']) == 0
        for nm in ref:
            assert np.abs(sol[nm] - ref[nm]).max() <= 1e-9, (k, nm)
        assert ('L' in sol) == latent


def test_sgl_batch_max_iter_status_and_mask():
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_SGL_batch
    p = 24
    S, _ = synth.make_problem("GGL", 1, p, N=60, seed=8)
    S = S[0]
    mask = np.ones((p, p))
    mask[:4, :] = mask[:, :4] = 0.3
    lams = np.array([0.07, 0.07, 0.07])
    res = ADMM_SGL_batch(S, lams, max_iter=6, tol=1e-20, rtol=1e-20, lambda1_mask=mask)
    ref, rinfo = orc.ADMM_SGL(S, 0.07, np.eye(p), max_iter=6, tol=1e-20, rtol=1e-20, lambda1_mask=mask)
    for sol, info in res:
        assert info['status'] == 'max iterations reached' and info['iterations'] == 6
        for nm in ref:
            assert np.abs(sol[nm] - ref[nm]).max() <= 1e-10


# ---- block_SGL (connected-component split on the host, equal-size blocks batched on the GPU) ------------

def test_g11_block_sgl(sol):
    g = load_golden("g11_block_sgl")
    S, lam = g["S"], float(g["lam"])
    p = S.shape[0]
    (out, text) = quiet(sol.block_SGL, S, lam, np.eye(p), tol=1e-10, rtol=1e-10)
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(out[nm] - g[nm]).max() <= 1e-9, nm
    assert text.count("ADMM terminated after") == int(np.sum(g["sizes"] > 1))
    assert np.abs(out["Theta"] - g["full_Theta"]).max() <= 1e-3          # reference tests/test_solvers.py:123-148
    (outm, _) = quiet(sol.block_SGL, S, lam, np.eye(p), tol=1e-10, rtol=1e-10, lambda1_mask=g["mask"])
    assert np.abs(outm["Theta"] - g["mask_Theta"]).max() <= 1e-9
    assert np.abs(outm["Omega"] - g["mask_Omega"]).max() <= 1e-9


def test_block_sgl_like_reference_test(sol):
    """reference tests/test_solvers.py:123-148: p=100 dense random S, lambda1=0.12, more than one component;
    block solution vs the un-split ADMM_SGL solution to 3 decimals -- here additionally vs the oracle."""
    rng = np.random.default_rng(0)
    p = 100
    A = rng.uniform(-1, 1, (p, p))
    S = A @ A.T / p
    S = 0.5 * (S + S.T) / np.sqrt(np.outer(np.diag(S), np.diag(S)))
    lam = 0.12 * 3
    numC, _ = sol.get_connected_components(S, lam)
    assert numC > 1
    (out, _) = quiet(sol.block_SGL, S, lam, np.eye(p), tol=1e-8, rtol=1e-8)
    ref = orc.block_SGL(S, lam, np.eye(p), tol=1e-8, rtol=1e-8)
    assert np.abs(out["Theta"] - ref["Theta"]).max() <= 1e-8
    ((full, _), _) = quiet(sol.ADMM_SGL, S, lam, np.eye(p), tol=1e-8, rtol=1e-8)
    assert np.abs(out["Theta"] - full["Theta"]).max() <= 1e-3


def test_block_sgl_ragged_components_in_padded_batches(sol):
    """Components of many different sizes -- below and above the LDS-Jacobi limit (p <= 128), so that both the Jacobi and
    the Newton-Schulz Omega-step see identity-padded slots -- each in the padded batch of its size class, with and without
    a lambda1_mask (one mask slice per instance, ggl_set_lambda1_mask_k): must equal the oracle's component-by-component
    loop (single_admm_solver.py:422-459), every component with its own iteration count."""
    from scipy.linalg import block_diag
    rng = np.random.default_rng(12)
    sizes = [2, 3, 3, 5, 8, 11, 14, 1, 30, 36, 40, 150, 170, 1]
    blocks = []
    for q in sizes:
        A = rng.standard_normal((q, 3 * q))
        blocks.append(A @ A.T / (3 * q) + 0.5 * np.eye(q))
    S = block_diag(*blocks)
    perm = rng.permutation(S.shape[0])
    S = S[np.ix_(perm, perm)]
    p = S.shape[0]
    lam = 0.02
    (out, text) = quiet(sol.block_SGL, S, lam, np.eye(p), tol=1e-9, rtol=1e-9)
    ref = orc.block_SGL(S, lam, np.eye(p), tol=1e-9, rtol=1e-9)
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(out[nm] - ref[nm]).max() <= 1e-9, nm
    assert text.count("ADMM terminated after") == sum(q > 1 for q in sizes)
    mask = rng.uniform(0.5, 1.5, (p, p))
    mask = 0.5 * (mask + mask.T)
    (outm, _) = quiet(sol.block_SGL, S, lam, np.eye(p), tol=1e-9, rtol=1e-9, lambda1_mask=mask)
    refm = orc.block_SGL(S, lam, np.eye(p), tol=1e-9, rtol=1e-9, lambda1_mask=mask)
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(outm[nm] - refm[nm]).max() <= 1e-9, nm


# ---- K-sharded driver on the real RCCL backend (single rank: exercises the device all-reduce path) -----------

def test_sharded_driver_on_rccl_single_rank(sol):
    """ADMM_MGL_sharded with backend nccl (= RCCL) and world size 1: the group sums of squares go through
    k_group_partial -> k_sum_chunks -> all_reduce on the ctx's device buffer (wrapped through
    __cuda_array_interface__) -> k_theta_ggl, on torch's current stream.  Must equal the unsharded solve."""
    import socket
    import torch
    import torch.distributed as dist
    from gglasso_amd import synth
    from gglasso_amd.dist import ADMM_MGL_sharded, TorchComm
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    import os
    try:
        # p = 150 takes the Newton-Schulz Omega-step, speculative after the first iteration: its validation flag rides
        # on the (p,p) all-reduce.  spec_factor 0.9 deflates the assumed bounds so that every speculative step
        # is rejected (by the all-reduced flag) and repeated; speculate 0 switches speculation off.
        for (K, p, env) in ((5, 40, {}), (3, 150, {}), (3, 150, {"spec_factor": 0.9}), (3, 150, {"speculate": 0}),
                            (4, 40, {"latent": 1}), (3, 150, {"latent": 1})):
            S, _ = synth.make_problem("GGL", K, p, seed=31)
            Om0 = np.stack([np.eye(p)] * K)
            comm = TorchComm(device="cuda:0")
            assert comm.stream_handle not in (None, 0)       # a dedicated stream, never the NULL handle
            kw = dict(tol=1e-9, rtol=1e-9, measure=True)
            if env.pop("latent", 0):      # latent variables under K-sharding: deferred norms as ONE row of local sums
                kw.update(latent=True, mu1=np.linspace(0.1, 0.2, K), max_iter=40)
            (a, ia), _ = quiet(ADMM_MGL_sharded, S, 0.05, 0.02, "GGL", Om0, K, comm, engine_kwargs={"options": env}, **kw)
            (b, ib), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.02, "GGL", Om0, **kw)
            assert ia["status"] == ib["status"]
            assert len(ia["residual"]) == len(ib["residual"])
            for nm in ("Omega", "Theta", "X", "L"):
                assert np.abs(a[nm] - b[nm]).max() <= 1e-10, (nm, env)
    finally:
        dist.destroy_process_group()


# ---- stress: extreme penalty parameters and scalings (condition numbers from 1 to 1e9 inside the Omega-step) ------

@pytest.mark.parametrize("rho0,scale,lam", [(1e-3, 1.0, 0.05), (1e3, 1.0, 0.05), (1.0, 1e3, 5.0), (1.0, 1e-3, 1e-4),
                                            (50.0, 30.0, 0.5)])
@pytest.mark.parametrize("reg,latent", [("GGL", False), ("FGL", True)])
def test_extreme_rho_and_scaling_vs_oracle(sol, rho0, scale, lam, reg, latent):
    """S scaled by 1e-3 .. 1e3 and rho0 from 1e-3 to 1e3: kappa(W^2 + 4/rho I) sweeps 1 .. 1e9, so both
    Newton-Schulz schedules, the L-step retry logic and the rho rule are all exercised; 15 iterations with
    rho updates must track the CPU oracle."""
    from gglasso_amd import synth
    K, p = 3, 140
    S, _ = synth.make_problem(reg, K, p, seed=23)
    S = S * scale
    Om0 = np.stack([np.eye(p)] * K)
    kw = dict(max_iter=15, tol=1e-20, rtol=1e-20, rho=rho0, latent=latent, mu1=0.2 * scale)
    ref, _ = orc.ADMM_MGL(S, lam, lam / 3, reg, Om0, **kw)
    (out, _), _ = quiet(sol.ADMM_MGL, S, lam, lam / 3, reg, Om0, **kw)
    for nm in ("Omega", "Theta", "L", "X"):
        tol = 1e-9 * max(1.0, np.abs(ref[nm]).max())
        assert np.abs(out[nm] - ref[nm]).max() <= tol, (nm, np.abs(out[nm] - ref[nm]).max())


def test_g12_batched_single_grid_search(sol):
    """The (lambda1[, mu1]) grid solved as one batch on the GPU against the reference's single_grid_search tables."""
    from grid_checks import check_single_grid_search, check_k_single_grid
    check_single_grid_search(load_golden)
    check_k_single_grid(load_golden)


@pytest.mark.parametrize("reg,K,p", [("GGL", 4, 160), ("FGL", 3, 150), ("GGL", 16, 200)])
def test_speculative_omega_step_hit_and_miss(sol, reg, K, p, monkeypatch):
    """The Omega-step runs its products on a schedule built from the previous iteration's spectral bounds and
    validates them on the device afterwards.  Hits (default 2 % inflation) and forced misses (bounds deflated by
    spec_factor 0.9: the Theta-step kernels must leave the iterate alone and the step is repeated) and no
    speculation at all must produce the same trajectory as the oracle."""
    from gglasso_amd import synth, solver
    S, _ = synth.make_problem(reg, K=K, p=p, N=2 * p, seed=31)
    Om0 = np.stack([np.eye(p)] * K)
    ref, _ = orc.ADMM_MGL(S, 0.05, 0.01, reg, Om0, max_iter=14, tol=1e-20, rtol=1e-20)
    stats = {}
    real_close = solver.HipEngine.close

    def closing(self):
        stats.update(self.ns_stats())
        real_close(self)

    monkeypatch.setattr(solver.HipEngine, "close", closing)
    for env, want_spec, want_miss in (({"speculate": 0}, False, False), ({}, True, False),
                                      ({"spec_factor": 0.9}, True, True)):
        monkeypatch.setattr(solver, "ENGINE_OPTIONS", dict(env))
        (s, info), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, reg, Om0, max_iter=14, tol=1e-20, rtol=1e-20)
        for nm in ('Omega', 'Theta', 'X'):
            assert np.abs(s[nm] - ref[nm]).max() <= 1e-9, (env, nm)
        assert (stats["spec_calls"] > 0) == want_spec, (env, stats)
        if not want_spec:
            assert stats["spec_misses"] == 0
        elif want_miss:
            assert stats["spec_misses"] == stats["spec_calls"], (env, stats)       # every attempt is rejected
        else:
            # from the identity start the spectrum still grows in the first iterations: a natural miss may occur
            assert stats["spec_misses"] < stats["spec_calls"], (env, stats)


# ---- batched lambda1 x lambda2 grid of the multiple-graph problems (G problems x K instances in one ctx) -----------

def test_g16_batched_mgl_grid_search(sol):
    """grid_search with the whole grid as one batch on the GPU against the reference's tables (GGL / FGL, AIC, w2,
    latent with ix_mu, thresholding, sequential mode, one-row grid)."""
    from grid_checks import check_mgl_grid_search
    quiet(check_mgl_grid_search, load_golden)


@pytest.mark.parametrize("reg,K,p,latent", [("GGL", 4, 40, False), ("GGL", 3, 150, False), ("FGL", 5, 140, True),
                                            ("GGL", 4, 200, True), ("FGL", 36, 36, False), ("GGL", 12, 60, False),
                                            ("GGL", 20, 50, True)])
def test_mgl_batch_equals_independent_solves(reg, K, p, latent):
    """Every problem of the batch must follow the trajectory of its own ADMM_MGL call (own rho, own stopping
    iteration) -- against the CPU oracle, below and above the LDS-Jacobi limit, both penalties, latent or not."""
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_MGL_batch
    S, _ = synth.make_problem(reg, K, p, N=2 * p, seed=41)
    Om0 = np.stack([np.eye(p)] * K)
    l1 = np.array([0.2, 0.08, 0.05, 0.03])
    l2 = np.array([0.05, 0.02, 0.03, 0.004])
    mu = np.linspace(0.2, 0.4, K) if latent else None
    res = ADMM_MGL_batch(S, l1, l2, reg, tol=1e-8, rtol=1e-8, latent=latent, mu1=mu, selection_stats=True)
    assert len(res) == 4
    for g in range(4):
        ref, rinfo = orc.ADMM_MGL(S, l1[g], l2[g], reg, Om0, tol=1e-8, rtol=1e-8, latent=latent, mu1=mu)
        out, info = res[g]
        assert info['status'] == rinfo['status'] == 'optimal'
        assert info['iterations'] == rinfo['iterations'], (g, info['iterations'], rinfo['iterations'])
        assert info['rho'] == rinfo['rho']
        for nm in ('Omega', 'Theta', 'L', 'X'):
            assert np.abs(out[nm] - ref[nm]).max() <= 1e-9, (g, nm)
        st = info['selection']
        assert np.allclose(st[:, 0], [np.sum(S[k] * ref['Theta'][k]) for k in range(K)], rtol=1e-9)
        assert np.allclose(st[:, 1], [np.linalg.slogdet(ref['Theta'][k])[1] for k in range(K)], rtol=1e-9, atol=1e-9)
        assert np.array_equal(st[:, 2], [np.count_nonzero(ref['Theta'][k]) for k in range(K)])


def test_mgl_batch_max_iter_and_nsamples():
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_MGL_batch
    K, p = 3, 30
    S, _ = synth.make_problem("GGL", K, p, N=90, seed=9)
    Om0 = np.stack([np.eye(p)] * K)
    nk = np.array([2.0, 3.0, 1.5])
    res = ADMM_MGL_batch(S, [0.05, 0.1], [0.02, 0.01], "GGL", max_iter=7, tol=1e-20, rtol=1e-20, n_samples=nk)
    for g, (l1, l2) in enumerate(((0.05, 0.02), (0.1, 0.01))):
        ref, _ = orc.ADMM_MGL(S, l1, l2, "GGL", Om0, max_iter=7, tol=1e-20, rtol=1e-20, n_samples=nk)
        out, info = res[g]
        assert info['status'] == 'max iterations reached' and info['iterations'] == 7
        for nm in ('Omega', 'Theta', 'X'):
            assert np.abs(out[nm] - ref[nm]).max() <= 1e-10


@pytest.mark.parametrize("K,p", [(4, 160), (16, 200), (6, 400)])
def test_fused_bound_partials_vs_norm_passes(sol, K, p, monkeypatch):
    """Spectral bounds from the epilogue partials of the B' launch (default) and from the separate norm passes over
    B' must drive the Omega-step to the same iterates (the bounds agree to rounding; both tracks follow the oracle)."""
    from gglasso_amd import synth
    S, _ = synth.make_problem("GGL", K=K, p=p, N=2 * p, seed=37)
    Om0 = np.stack([np.eye(p)] * K)
    ref, _ = orc.ADMM_MGL(S, 0.05, 0.01, "GGL", Om0, max_iter=8, tol=1e-20, rtol=1e-20)
    outs = []
    for fused in (1, 0):
        monkeypatch.setattr(sol, "ENGINE_OPTIONS", {"fused_bounds": fused})
        (s, _), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, "GGL", Om0, max_iter=8, tol=1e-20, rtol=1e-20)
        for nm in ('Omega', 'Theta', 'X'):
            assert np.abs(s[nm] - ref[nm]).max() <= 1e-9, (fused, nm)
        outs.append(s)
    assert np.abs(outs[0]['Omega'] - outs[1]['Omega']).max() <= 1e-11


def test_sharded_driver_rccl_behind_the_c_abi_single_rank(sol):
    """ggl_comm_init / ggl_admm_step_sharded: RCCL resolved by the library itself (dlopen), both all-reduces issued in C
    on the ctx stream, one call per iteration; world size 1 (the 1-GPU box), hits, forced misses and no speculation.
    torch.distributed (gloo) only ships the unique id."""
    import socket
    import torch.distributed as dist
    from gglasso_amd import synth
    from gglasso_amd.dist import ADMM_MGL_sharded, RcclComm
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        # (8, 500): a slab that runs as TWO concurrent parts inside ggl_admm_step_sharded (the 4-GPU slab of the headline)
        for (K, p, env) in ((5, 40, {}), (3, 150, {}), (3, 150, {"spec_factor": 0.9}), (3, 150, {"speculate": 0}), (4, 500, {}),
                            (8, 500, {}), (8, 500, {"spec_factor": 0.9}), (4, 40, {"latent": 1}), (3, 150, {"latent": 1})):
            S, _ = synth.make_problem("GGL", K, p, seed=31)
            Om0 = np.stack([np.eye(p)] * K)
            kw = dict(tol=1e-9, rtol=1e-9) if p < 500 else dict(tol=1e-20, rtol=1e-20, max_iter=8)
            if env.pop("latent", 0):
                # K-sharded run with latent variables (ggl_admm_step_sharded_latent): L-step on the slab, per-instance mu1
                kw.update(latent=True, mu1=np.linspace(0.1, 0.2, K), max_iter=40)
            (a, ia), _ = quiet(ADMM_MGL_sharded, S, 0.05, 0.02, "GGL", Om0, K, RcclComm(), measure=True,
                               engine_kwargs={"options": env}, **kw)
            (b, ib), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.02, "GGL", Om0, measure=True, **kw)
            assert ia["status"] == ib["status"]
            assert len(ia["residual"]) == len(ib["residual"])
            for nm in ("Omega", "Theta", "X", "L"):
                assert np.abs(a[nm] - b[nm]).max() <= 1e-10, (nm, env)
        # a dual start that is symmetric only to ~1e-8: the mirroring tile-pair Theta kernel reads the upper triangle of
        # the all-reduced FULL group-sum matrix (the per-element kernels would read both triangles)
        K, p = 4, 150
        S, _ = synth.make_problem("GGL", K, p, seed=33)
        Om0 = np.stack([np.eye(p)] * K)
        rng = np.random.default_rng(6)
        X0 = 0.01 * rng.standard_normal((K, p, p))
        X0 = 0.5 * (X0 + X0.transpose(0, 2, 1)) + 1e-8 * np.triu(rng.standard_normal((K, p, p)), 1)
        (a, _), _ = quiet(ADMM_MGL_sharded, S, 0.05, 0.02, "GGL", Om0, K, RcclComm(), X_0=X0, max_iter=8, tol=1e-20, rtol=1e-20)
        (b, _), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.02, "GGL", Om0, X_0=X0, max_iter=8, tol=1e-20, rtol=1e-20)
        ref, _ = orc.ADMM_MGL(S, 0.05, 0.02, "GGL", Om0, X_0=X0, max_iter=8, tol=1e-20, rtol=1e-20)
        for nm in ("Omega", "Theta", "X"):
            assert np.abs(a[nm] - b[nm]).max() <= 1e-10, nm
            assert np.abs(a[nm] - ref[nm]).max() <= 1e-9, nm
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("early", [0, 1])
@pytest.mark.parametrize("reg,K,p", [("GGL", 4, 160), ("GGL", 16, 200), ("FGL", 3, 150), ("GGL", 8, 400)])
def test_pipelined_iterations_are_bitwise_the_unpipelined_ones(sol, reg, K, p, early, monkeypatch):
    """GGL_OPT_PIPELINE: ggl_admm_step launches the next iteration's Omega-step chain before it returns.  The chain only
    touches scratch and Omega[cur^1], so the iterates must be BITWISE those of the unpipelined run -- with rho changes
    (chains dropped), with the state read in the middle of the loop (chain dropped by ggl_get_state), and at the end.
    GGL_OPT_EARLY_PART additionally puts the first part of the chain (W, A', B') into the stream BEFORE the wait for the
    residuals, with a schedule built from the bounds validated one iteration earlier: same iteration, the Omega-step's
    coefficients differ in the last digits -- iterates within 1e-10 of the unpipelined ones instead of bitwise."""
    from gglasso_amd import synth, solver
    S, _ = synth.make_problem(reg, K=K, p=p, N=2 * p, seed=43)
    Om0 = np.stack([np.eye(p)] * K)
    outs, stats, pipe_stats = [], [], []
    n_it = 16 if not early else 40
    # (third run, early part only: GGL_OPT_FUSED_W off -- the W of an early first part comes from k_form_W_sym instead of the
    # Theta kernel that precedes it: the same arithmetic per element, the same iterates)
    # (fourth run, early part only: the riders off -- bound validation, table transfers and norm reduction as launches of their
    # own instead of extra workgroups of a product launch: the same arithmetic in the same order, bitwise the second run)
    no_riders = {"cw_rider": 0, "copy_rider": 0, "reduce_rider": 0}
    for pipe, fused_w, extra in ((0, 1, {}), (1, 1, {})) + (((1, 0, {}), (1, 1, no_riders)) if early else ()):
        eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S),
                               options={"pipeline": pipe, "early_part": early, "fused_w": fused_w, **extra})
        nk = np.ones(K)
        rho, mid = 1.0, None
        for it in range(n_it):
            sq = eng.step(rho, 0.05, 0.01, reg, False, None, nk).copy()
            r_t, s_t, _, _ = solver.residuals_from_norms(sq, rho, 1e-20, 1e-20, 1.0)
            new = solver.next_rho(rho, r_t, s_t)
            if new != rho:
                eng.scale_X(rho / new)
            rho = new
            if it == 9:
                mid = eng.state()          # any entry point other than the step drops a pre-launched chain first
        outs.append((mid, eng.state(), rho))
        stats.append(eng.ns_stats())
        pipe_stats.append(eng.pipeline_stats())
        eng.close()
    for a, b in zip(outs[0][:2], outs[1][:2]):
        for nm in ("Omega", "Theta", "X"):
            if early:
                assert np.abs(a[nm] - b[nm]).max() <= 1e-10, nm
                assert np.array_equal(b[nm], b[nm].transpose(0, 2, 1)), nm
            else:
                assert np.array_equal(a[nm], b[nm]), nm
    assert outs[0][2] == outs[1][2]
    assert stats[0]["pre_dropped"] == 0 and stats[1]["spec_calls"] > stats[0]["spec_calls"]
    assert stats[1]["pre_dropped"] >= 1                      # the chain behind iteration 10 was dropped by state()
    assert pipe_stats[0]["early_launched"] == 0
    if early:
        assert pipe_stats[1]["early_launched"] >= 5 and pipe_stats[1]["early_used"] >= pipe_stats[1]["early_launched"] - 3, pipe_stats[1]
        if reg == "GGL":
            assert pipe_stats[1]["w_fused_used"] >= pipe_stats[1]["early_launched"] - 1, pipe_stats[1]
        assert pipe_stats[2]["w_fused"] == 0 and pipe_stats[2]["early_launched"] == pipe_stats[1]["early_launched"]
        for a, b in zip(outs[1][:2], outs[2][:2]):
            for nm in ("Omega", "Theta", "X"):
                assert np.abs(a[nm] - b[nm]).max() <= 1e-13, nm
        for a, b in zip(outs[1][:2], outs[3][:2]):
            for nm in ("Omega", "Theta", "X"):
                assert np.array_equal(a[nm], b[nm]), nm
        assert stats[1]["spec_misses"] == stats[3]["spec_misses"]
        assert pipe_stats[3]["bound_rides"] == pipe_stats[3]["copy_rides"] == pipe_stats[3]["reduce_rides"] == 0
        assert pipe_stats[1]["bound_rides"] >= 5, pipe_stats[1]
        if reg == "GGL":
            assert pipe_stats[1]["reduce_rides"] >= 5, pipe_stats[1]
            if (K, p) != (8, 400):                                # (one launch sequence: the tables ride as well)
                assert pipe_stats[1]["copy_rides"] >= 5, pipe_stats[1]
    else:
        assert pipe_stats[1]["early_launched"] == 0
    ref, _ = orc.ADMM_MGL(S, 0.05, 0.01, reg, Om0, max_iter=n_it, tol=1e-20, rtol=1e-20)
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(outs[1][1][nm] - ref[nm]).max() <= 1e-9, nm


def test_fgl_K_beyond_the_scan_buffer_is_refused_with_a_message(sol):
    """FGL with more instances than fit one workgroup's LDS scan buffer (Condat's scan is serial
    along K) used to surface as a bare hipErrorInvalidValue; it is an argument error with a message now."""
    from gglasso_amd import ops
    K, p = 330, 6
    S = np.stack([np.eye(p)] * K)
    with pytest.raises(AssertionError, match="exceeds the 318 instances"):
        quiet(sol.ADMM_MGL, S, 0.05, 0.01, "FGL", S.copy(), max_iter=2)
    with pytest.raises(AssertionError, match="exceeds the 318 instances"):
        ops.prox_p(S, 0.1, 0.1, "FGL")
    (s, info), _ = quiet(sol.ADMM_MGL, S[:318], 0.05, 0.01, "FGL", S[:318].copy(), max_iter=2)      # the limit itself works
    assert info["status"] in ("optimal", "max iterations reached") and np.isfinite(s["Theta"]).all()


def test_omega_step_tolerance_and_carried_bound_vector(sol):
    """GGL_OPT_NS_TOL / GGL_OPT_CW_WARM.  The Omega-step's square-root iteration stops at a relative spectral accuracy
    (default 2e-12; 0 = fp64 resolution): the default must never cost more products than the exact mode, the exact mode
    must reproduce the reference's eigendecomposition route to ~1e-12 and the default must stay orders of magnitude
    inside the 1e-8 of the reference comparison.  The Collatz-Wielandt vector carried across iterations only tightens a
    bound that is rigorous for any positive vector: same iterates to rounding, no speculation miss, no more work."""
    from gglasso_amd import synth, solver
    K, p = 6, 384
    S, _ = synth.make_problem("GGL", K=K, p=p, N=2 * p, seed=47)
    Om0 = np.stack([np.eye(p)] * K)
    ref, _ = orc.ADMM_MGL(S, 0.05, 0.01, "GGL", Om0, max_iter=14, tol=1e-20, rtol=1e-20, update_rho=False)
    res = {}
    for name, opts in (("default", {}), ("exact", {"ns_tol": 0.0}), ("cold", {"cw_warm": 0}), ("loose", {"ns_tol": 1e-9})):
        eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S), options=opts)
        for it in range(14):
            eng.step(1.0, 0.05, 0.01, "GGL", False, None, np.ones(K))
        res[name] = (eng.state(), eng.ns_stats(), eng.get_option("ns_tol"))
        eng.close()
    assert res["default"][2] == 2e-12 and res["exact"][2] == 4e-16
    err = {n: max(np.abs(res[n][0][nm] - ref[nm]).max() for nm in ("Omega", "Theta", "X")) for n in res}
    assert err["exact"] <= 2e-12, err
    assert err["default"] <= 1e-10 and err["cold"] <= 1e-10, err
    assert err["loose"] <= 1e-6, err
    units = {n: res[n][1]["units"] for n in res}
    assert units["loose"] <= units["default"] <= units["exact"], units
    assert units["default"] <= units["cold"], units
    for n in res:
        assert res[n][1]["spec_misses"] == 0 and res[n][1]["eigh_fallbacks"] == 0, (n, res[n][1])
    with pytest.raises(AssertionError):
        solver.HipEngine(S, Om0, Om0, np.zeros_like(S), options={"ns_tol": 1e-3})


@pytest.mark.parametrize("K,p", [(48, 40), (130, 28)])
def test_large_K_latent_and_fgl_free_paths_of_the_wide_theta_kernel(sol, K, p):
    """K > 32 through the per-element Theta kernel without the fused dual update (latent: Theta and C = Theta - X - Omega
    are written, the dual update follows the L-step), against the oracle (admm_solver.py:190-208)."""
    from gglasso_amd import synth
    S, _ = synth.make_problem("GGL", K=K, p=p, N=2 * p, seed=41)
    Om0 = np.stack([np.eye(p)] * K)
    kw = dict(max_iter=6, tol=1e-20, rtol=1e-20, latent=True, mu1=0.2)
    ref, _ = orc.ADMM_MGL(S, 0.05, 0.01, "GGL", Om0, **kw)
    (s, _), _ = quiet(sol.ADMM_MGL, S, 0.05, 0.01, "GGL", Om0, **kw)
    for nm in ('Omega', 'Theta', 'L', 'X'):
        assert np.abs(s[nm] - ref[nm]).max() <= 1e-9, nm


def test_state_snapshot_restore_equals_a_fresh_upload(sol):
    """ggl_state_snapshot: a solve restarted from the device copy of its start point runs the iterations a solve from the
    re-uploaded start point runs -- bit for bit (everything carried across iterations is forgotten both ways)."""
    from gglasso_amd import synth
    K, p = 6, 180
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=3)
    Om0 = np.stack([np.eye(p)] * K)
    X0 = np.zeros_like(S)
    nk = np.ones(K)
    eng = sol.HipEngine(S, Om0, Om0, X0)
    try:
        eng.save_state()
        outs = []
        for mode in ("first", "restore", "upload"):
            if mode == "restore":
                eng.restore_state()
            elif mode == "upload":
                eng.set_state(Om0, Om0, X0)
            for _ in range(7):
                eng.step(1.3, 0.05, 0.01, "GGL", False, None, nk)
            outs.append(eng.state())
    finally:
        eng.close()
    for nm in ("Omega", "Theta", "X"):
        assert np.array_equal(outs[0][nm], outs[1][nm]), nm
        assert np.array_equal(outs[0][nm], outs[2][nm]), nm


def test_shared_start_arrays_are_replicated_on_the_device():
    """ggl_set_S_ex / ggl_set_state_ex: a (K,p,p) broadcast VIEW of one (p,p) matrix (what the batched grids pass for S,
    Omega_0, X_0) is uploaded once and replicated on the device -- the engine's state must equal the materialised stacks',
    for K a power of two and not, with and without L."""
    from gglasso_amd import solver
    rng = np.random.default_rng(12)
    p = 70
    A = [rng.standard_normal((p, p)) for _ in range(5)]
    A = [0.5 * (a + a.T) for a in A]
    for K in (1, 2, 5, 8, 13):
        views = [np.broadcast_to(a, (K, p, p)) for a in A]
        full = [np.ascontiguousarray(v) for v in views]
        for with_L in (False, True):
            e1 = solver.HipEngine(views[0], views[1], views[2], views[3], L_0=views[4] if with_L else None)
            e2 = solver.HipEngine(full[0], full[1], full[2], full[3], L_0=full[4] if with_L else None)
            try:
                s1, s2 = e1.state(), e2.state()
                for nm in ("Omega", "Theta", "X", "L"):
                    assert np.array_equal(s1[nm], s2[nm]), (K, with_L, nm)
                assert np.array_equal(s1["Omega"], full[1]) and np.array_equal(s1["X"], full[3])
                # mixed: one shared, one per-instance array
                per = np.ascontiguousarray(full[2] + np.arange(K)[:, None, None])
                e1.close()
                e1 = solver.HipEngine(views[0], views[1], per, views[3])
                assert np.array_equal(e1.state()["Theta"], per) and np.array_equal(e1.state()["Omega"], full[1])
            finally:
                e1.close()
                e2.close()
    # (G,K',p,p) views: the stack of ONE multiple-graph problem for each of G grid points (ADMM_MGL_batch)
    Kp = 3
    B = [np.stack([a + k for k in range(Kp)]) for a in A[:4]]
    for G in (2, 5, 8):
        views = [np.broadcast_to(b, (G, Kp, p, p)) for b in B]
        full = [np.ascontiguousarray(v).reshape(G * Kp, p, p) for v in views]
        zero = np.broadcast_to(np.zeros((p, p)), (G * Kp, p, p))
        e1 = solver.HipEngine(views[0], views[1], views[2], zero)
        e2 = solver.HipEngine(full[0], full[1], full[2], np.zeros((G * Kp, p, p)))
        try:
            assert e1.K == G * Kp
            s1, s2 = e1.state(), e2.state()
            for nm in ("Omega", "Theta", "X", "L"):
                assert np.array_equal(s1[nm], s2[nm]), (G, nm)
            assert np.array_equal(s1["Theta"], full[2])
        finally:
            e1.close()
            e2.close()


@pytest.mark.parametrize("K,p", [(4, 300), (8, 400), (20, 200)])
def test_bound_validation_on_a_side_stream_is_bitwise(sol, dev_library, K, p):
    """GGL_OPT_BOUND_SIDE: the kernels that validate a speculative Omega-step's assumed bound run on a side stream beside the
    chain's first products and are joined before B' is overwritten -- the same kernels on the same data, so the iterates
    are bitwise those of the in-chain order (one part, two concurrent parts at K = 8, a three-step schedule at (20,200));
    with spec_factor 0.9 every speculative step is REJECTED by those kernels and repeated: the flag must arrive in time."""
    from gglasso_amd import synth, solver
    S, _ = synth.make_problem("GGL", K=K, p=p, N=2 * p, seed=7)
    Om0 = np.stack([np.eye(p)] * K)
    nk = np.ones(K)
    for extra in ({}, {"spec_factor": 0.9}):
        outs = []
        # (side, join_flag): the validation kernels in the chain / on a side stream; the parts joined through flag words in
        # device memory (GGL_OPT_JOIN_FLAG, the default) / through a cross-queue event -- all four orders of the same kernels
        # rider: the validation as extra workgroups of the first product launch behind B' (GGL_OPT_CW_RIDER, the default; same
        # arithmetic in the same order) / as the two kernels / (2) as the launch of its own a rider gets that no product takes
        rides = []
        # copy rider (GGL_OPT_COPY_RIDER, the default): the chain's table transfers as extra workgroups of the A' launch / as
        # the copy kernel in front of it
        copies = []
        for side, jf, rider, cpr in ((0, 1, 0, 0), (1, 1, 0, 0), (0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 1, 0), (0, 0, 1, 0), (0, 1, 2, 0),
                                     (0, 1, 1, 2), (0, 1, 0, 2)):
            # (the reduce rider -- GGL_OPT_REDUCE_RIDER, the norm reduction in the next chain's A' launch -- goes with the copy
            # rider here; test_pipelined_iterations_are_bitwise_the_unpipelined_ones has the early parts it needs)
            eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S),
                                   options={"bound_side": side, "join_flag": jf, "cw_rider": rider, "copy_rider": cpr,
                                            "reduce_rider": cpr, **extra})
            rho = 1.0
            for it in range(14):
                sq = eng.step(rho, 0.05, 0.01, "GGL", False, None, nk).copy()
                r_t, s_t, _, _ = solver.residuals_from_norms(sq, rho, 1e-20, 1e-20, 1.0)
                new = solver.next_rho(rho, r_t, s_t)
                if new != rho:
                    eng.scale_X(rho / new)
                rho = new
            outs.append((eng.state(), eng.ns_stats()))
            rides.append(eng.pipeline_stats()["bound_rides"])
            copies.append(eng.pipeline_stats()["copy_rides"])
            eng.close()
        for o in outs[1:]:
            for nm in ("Omega", "Theta", "X"):
                assert np.array_equal(outs[0][0][nm], o[0][nm]), (nm, extra)
        misses = [o[1]["spec_misses"] for o in outs]
        assert len(set(misses)) == 1, (misses, rides, extra)
        assert rides[:4] == [0, 0, 0, 0] and rides[8] == 0 and min(rides[4:8]) >= (2 if extra else 6), rides   # (a rejected step leaves no vector to ride on)
        assert copies[:7] == [0] * 7 and min(copies[7:]) >= (1 if extra else 6), copies
        if extra:
            assert outs[1][1]["spec_misses"] >= 2
        else:
            assert outs[1][1]["spec_calls"] >= 8


def test_download_touches_fresh_pages_first_and_returns_the_same_bytes(sol):
    """GGL_OPT_DOWNLOAD_THREADS: a whole-state download of more than 32 MB touches the pages of the caller's arrays from several
    host threads before the copy (tools/time_download.py) -- the arrays must come back with the same bytes as with one thread,
    into fresh arrays and into arrays that hold other data."""
    from gglasso_amd import synth, solver
    K, p = 6, 600                                  # 4 stacks of 17 MB
    S, _ = synth.make_problem("GGL", K=K, p=p, N=2 * p, seed=11)
    Om0 = np.stack([np.eye(p)] * K)
    eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S))
    try:
        for _ in range(3):
            eng.step(1.0, 0.05, 0.01, "GGL", False, None, np.ones(K))
        eng.set_option("download_threads", 1)
        one = eng.state()
        eng.set_option("download_threads", 8)
        many = eng.state()
        for nm in one:
            assert np.array_equal(one[nm], many[nm]), nm
        assert np.abs(many["Theta"]).max() > 0 and np.isfinite(many["Omega"]).all()
    finally:
        eng.close()
