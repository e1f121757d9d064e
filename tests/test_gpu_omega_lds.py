"""The Omega-step of small matrices (p <= 64) as ONE launch with the Newton-Schulz chain resident in LDS (csrc/omega_lds.hip,
GGL_OPT_OMEGA_LDS): the kernel against numpy.linalg.eigh's phiplus (solver/ggl_helper.py:272-303), its range check, and the
solver paths that launch it -- unvalidated where the caller can repeat a step (ggl_admm_step, the batched steps), behind a
stream synchronisation elsewhere (latent models, ggl_step_omega) -- with the step repeated on the launch chain when an
instance falls outside the kernel's range.
"""
import contextlib
import io

import numpy as np
import pytest

from oracle import ggl_oracle as orc

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _sym(rng, K, p, scale):
    A = rng.standard_normal((K, p, p)) * scale / np.sqrt(p)
    return np.ascontiguousarray(0.5 * (A + A.transpose(0, 2, 1)))


def _dev_omega(Theta, X, S, beta, L=None, tol=2e-12, degrees=9):
    from gglasso_amd import _lib
    from gglasso_amd._lib import ptr
    lib = _lib.load()
    K, p, _ = Theta.shape
    Om, cb, out = np.zeros_like(Theta), np.zeros(K), np.zeros(18)
    _lib.check(lib.ggl_dev_omega_lds(K, p, ptr(Theta), ptr(L) if L is not None else None, ptr(X), ptr(S), ptr(beta), tol,
                                     degrees, ptr(Om), ptr(cb), 1, ptr(out)))
    return Om, cb, out


@pytest.mark.parametrize("p", [1, 2, 9, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64])
@pytest.mark.parametrize("latent", [False, True])
def test_kernel_against_eigh(p, latent):
    """Every padded tile size (16 / 32 / 48 / 64) at its edges, odd and even p; W is formed in the kernel from the iterate
    (Theta - L - X - beta S, admm_solver.py:180-187) and only its LOWER triangle is read, as numpy.linalg.eigh does."""
    rng = np.random.default_rng(1000 + p)
    K = 5
    Theta, X, S = _sym(rng, K, p, 1.0) + np.eye(p), _sym(rng, K, p, 0.3), _sym(rng, K, p, 1.0)
    L = _sym(rng, K, p, 0.2) if latent else None
    beta = rng.uniform(0.3, 2.0, K)
    # garbage above the diagonal of every input must not matter
    dirty = [a + np.triu(rng.standard_normal((K, p, p)), 1) for a in (Theta, X, S)] + ([L + np.triu(rng.standard_normal((K, p, p)), 1)] if latent else [])
    W = Theta - (L if latent else 0.0) - X - beta[:, None, None] * S
    ref, _ = orc.phiplus_stack(W, beta)
    lam = np.array([np.linalg.eigvalsh(W[k] @ W[k] + 4 * beta[k] * np.eye(p))[-1] for k in range(K)])
    first = {}
    # waves per workgroup: 0 = what the solver runs (two per SIMD above p = 32, round 5), 4 and 8 explicitly -- the blocks
    # are dealt differently, every block's arithmetic is the same: bit-identical results
    for waves in (0, 4, 8):
        for tol, bar in ((2e-12, 1e-11), (1e-10, 5e-10)):
            Om, cb, out = _dev_omega(*[np.ascontiguousarray(a) for a in dirty[:3]], beta,
                                     L=np.ascontiguousarray(dirty[3]) if latent else None, tol=tol, degrees=9 + 1000 * waves)
            assert out[1] == 0
            assert np.abs(Om - ref).max() <= bar * max(1.0, np.abs(ref).max()), (p, tol, waves)
            assert np.array_equal(Om, Om.transpose(0, 2, 1))
            assert np.all(cb >= lam * (1 - 1e-12)) and np.all(cb <= 4.0 * lam + 1e-300)      # a bound, and not a wild one
            assert np.array_equal(first.setdefault(tol, Om), Om), (p, tol, waves)


@pytest.mark.parametrize("degrees", [3, 5, 9])
def test_kernel_over_its_range_of_condition_numbers(degrees):
    """kappa(W^2 + 4 beta I) from 1 to just under 300, every degree set of the schedule table (six-step schedules at the
    most: beyond them, and beyond kappa = 300, the kernel raises its flag instead of computing)."""
    rng = np.random.default_rng(23)
    p, beta = 40, 0.7
    served = 0
    for kappa in (1.0, 1.3, 4.0, 30.0, 120.0, 280.0):
        wmax = np.sqrt(4 * beta * (kappa - 1.0))
        W = np.empty((3, p, p))
        for k in range(3):
            Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
            w = rng.uniform(-wmax, wmax, p)
            w[0], w[1] = wmax, -wmax
            W[k] = (Q * w) @ Q.T
            W[k] = 0.5 * (W[k] + W[k].T)
        Z = np.zeros_like(W)
        ref, _ = orc.phiplus_stack(W, beta)
        Om, cb, out = _dev_omega(W, Z, Z, np.full(3, beta), degrees=degrees)
        if out[1]:
            assert degrees < 9 or kappa > 100.0, (degrees, kappa)     # (cubic / quintic schedules run out of steps earlier)
            continue
        served += 1
        assert np.abs(Om - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (degrees, kappa)
    assert served >= {9: 5, 5: 4, 3: 3}[degrees]     # (the kernel's own bound overestimates lambda_max by up to ~1.3: kappa = 280 may read as > 300)


def test_kernel_flags_what_it_cannot_do():
    rng = np.random.default_rng(5)
    p = 24
    W = _sym(rng, 4, p, 1.0)
    Z = np.zeros_like(W)
    beta = np.ones(4)
    assert _dev_omega(W, Z, Z, beta)[2][1] == 0
    big = W.copy()
    big[2] *= 100.0                                  # kappa ~ 1e4 in one instance: the flag, whatever the others do
    assert _dev_omega(big, Z, Z, beta)[2][1] == 1
    nan = W.copy()
    nan[1, 3, 2] = np.nan
    assert _dev_omega(nan, Z, Z, beta)[2][1] == 1
    with pytest.raises(Exception, match="range"):
        _dev_omega(_sym(rng, 1, 65, 1.0), np.zeros((1, 65, 65)), np.zeros((1, 65, 65)), np.ones(1))


def _ggl(K, p, seed):
    from gglasso_amd import synth
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=seed)
    return S, np.repeat(np.eye(p)[None], K, axis=0)


@pytest.fixture()
def engines(monkeypatch):
    """lds_stats / ns_stats / last_dispatch of every engine a solver call creates, captured when it is closed."""
    from gglasso_amd import solver
    seen = []
    real_close = solver.HipEngine.close

    def closing(self):
        if getattr(self, "h", None):
            seen.append({"lds": self.lds_stats(), "ns": self.ns_stats(), "dispatch": self.last_dispatch()})
        real_close(self)

    monkeypatch.setattr(solver.HipEngine, "close", closing)
    return seen


@pytest.mark.parametrize("reg,K,p,latent", [("GGL", 6, 40, False), ("FGL", 4, 33, False), ("GGL", 3, 64, True)])
def test_admm_mgl_runs_on_the_lds_kernel_and_matches_the_oracle(engines, reg, K, p, latent):
    from gglasso_amd import solver
    from oracle_engine import OracleEngine
    S, Om0 = _ggl(K, p, 77 + p)
    kw = dict(tol=1e-9, rtol=1e-9, max_iter=60, latent=latent, mu1=(0.4 * np.ones(K) if latent else None))
    sol, info = quiet(solver.ADMM_MGL, S, 0.05, 0.02, reg, Om0, **kw)
    st = engines[-1]
    assert latent or st["dispatch"]["variant"] == 41                 # (latent: the L-step's product kernel ran last)
    assert st["lds"]["calls"] >= 10 and st["lds"]["misses"] == 0
    assert st["lds"]["calls"] == st["ns"]["calls"]                   # every Omega-step of the solve
    assert 6 * K * st["lds"]["calls"] <= st["lds"]["products"] <= 13 * K * st["lds"]["calls"]
    real = solver.ENGINE
    solver.ENGINE = OracleEngine
    try:
        ref, rinfo = quiet(solver.ADMM_MGL, S, 0.05, 0.02, reg, Om0, **kw)
    finally:
        solver.ENGINE = real
    assert info["status"] == rinfo["status"]
    for key in ("Omega", "Theta", "X") + (("L",) if latent else ()):
        assert np.linalg.norm(sol[key] - ref[key]) <= 1e-8, key
    # and the same solve without the kernel: the launch chain's schedule is shared by the batch, the kernel's is per instance --
    # the two agree to the iteration's own accuracy, not bitwise
    solver.ENGINE_OPTIONS["omega_lds"] = 0
    try:
        sol0, _ = quiet(solver.ADMM_MGL, S, 0.05, 0.02, reg, Om0, **kw)
    finally:
        solver.ENGINE_OPTIONS.pop("omega_lds")
    assert engines[-1]["dispatch"]["variant"] != 41 and engines[-1]["lds"]["calls"] == 0 and engines[-1]["ns"]["calls"] >= 10
    assert np.linalg.norm(sol["Theta"] - sol0["Theta"]) <= 1e-8


def test_an_instance_outside_the_range_repeats_the_step_on_the_launch_chain(engines):
    """A badly scaled problem: kappa(W^2 + 4 beta I) > 300 from the first step on.  The kernel raises its flag, the Theta-step
    leaves the iterate alone, the step is repeated on the launch chain and the kernel sits out a growing number of steps."""
    from gglasso_amd import solver
    K, p = 4, 32
    S, Om0 = _ggl(K, p, 3)
    S = S * 60.0
    kw = dict(tol=1e-9, rtol=1e-9, max_iter=40, update_rho=False)
    sol, info = quiet(solver.ADMM_MGL, S, 0.5, 0.2, "GGL", Om0, **kw)
    st = engines[-1]
    assert st["lds"]["misses"] >= 1
    assert st["lds"]["calls"] < st["ns"]["calls"]                    # the launch chain took the repeated and the sat-out steps
    solver.ENGINE_OPTIONS["omega_lds"] = 0
    try:
        sol0, info0 = quiet(solver.ADMM_MGL, S, 0.5, 0.2, "GGL", Om0, **kw)
    finally:
        solver.ENGINE_OPTIONS.pop("omega_lds")
    assert info["status"] == info0["status"]
    scale = max(1.0, np.abs(sol0["Theta"]).max())
    assert np.abs(sol["Theta"] - sol0["Theta"]).max() <= 1e-9 * scale
    assert np.abs(sol["Omega"] - sol0["Omega"]).max() <= 1e-9 * scale


def test_batched_sgl_path_unvalidated_launch_and_repeat(engines):
    """ggl_sgl_batch_step at p <= 64 (round 5): ONE launch per iteration -- the workgroup that holds an instance's Omega in LDS
    goes on with its Theta-step, dual update and stopping-test sums (k_omega_lds<.., SGL>).  A badly scaled point of the
    batch marks itself and leaves its iterate alone; it is redone ALONE on the launch chain (a compact ctx of the marked
    instances, scattered back), the others' step stands, and the kernel sits out the next steps as after any miss."""
    from gglasso_amd import batch, solver
    p = 30
    S, _ = _ggl(1, p, 11)
    lam = np.array([0.05, 0.1, 0.2, 0.4])
    res = quiet(batch.ADMM_SGL_batch, S[0], lam, tol=1e-9, rtol=1e-9, max_iter=200)
    assert engines[-1]["lds"]["calls"] >= 10 and engines[-1]["lds"]["misses"] == 0
    for k, l in enumerate(lam):
        one, _ = quiet(solver.ADMM_SGL, S[0], float(l), np.eye(p), tol=1e-9, rtol=1e-9, max_iter=200)
        assert np.abs(res[k][0]["Theta"] - one["Theta"]).max() <= 1e-8
    # per-instance covariance matrices, one of them scaled out of the kernel's range
    Sk = np.repeat(S, 3, axis=0)
    Sk[1] *= 80.0
    n0 = len(engines)
    resk = quiet(batch.ADMM_SGL_batch, Sk, np.array([0.1, 0.1, 0.1]), tol=1e-9, rtol=1e-9, max_iter=300, compact=False)
    assert sum(e["lds"]["misses"] for e in engines[n0:]) >= 1
    one, _ = quiet(solver.ADMM_SGL, S[0], 0.1, np.eye(p), tol=1e-9, rtol=1e-9, max_iter=300)
    assert np.abs(resk[0][0]["Theta"] - one["Theta"]).max() <= 1e-8
    assert np.abs(resk[2][0]["Theta"] - one["Theta"]).max() <= 1e-8
    bad, _ = quiet(solver.ADMM_SGL, Sk[1], 0.1, np.eye(p), tol=1e-9, rtol=1e-9, max_iter=300)
    assert np.abs(resk[1][0]["Theta"] - bad["Theta"]).max() <= 1e-8 * max(1.0, np.abs(bad["Theta"]).max())


def test_random_small_batches_of_single_problems_against_the_oracle():
    """The one-launch iteration of a batch of single problems (k_omega_lds<.., SGL>: Omega-step, prox_od_1norm, dual update and
    stopping-test sums by the workgroup that holds the instance; single_admm_solver.py:157-214) on random p <= 64, K, lambda1,
    with a shared mask, per-instance masks, padded instances of different dimension (block_SGL's batches) and a badly scaled
    point (redone alone on the launch chain): Theta and the exit status of every point against the oracle's ADMM_SGL.
    tools/fuzz_sgl_batch.py is the long form."""
    from gglasso_amd import synth
    from gglasso_amd.batch import ADMM_SGL_batch, pad_blocks
    rng = np.random.default_rng(3)
    kw = dict(tol=1e-8, rtol=1e-7, max_iter=400)
    for case, kind in enumerate(["plain", "mask", "maskK", "dims", "scaled", "plain", "dims", "scaled"]):
        p, K = int(rng.integers(3, 65)), int(rng.integers(2, 9))
        S, _ = synth.make_problem("GGL", K, p, N=2 * p + 5, seed=100 + case)
        lam = np.exp(rng.uniform(np.log(0.03), np.log(0.5), K))
        ref_S, ref_mask, args = [S[k] for k in range(K)], [None] * K, dict(kw)
        if kind == "mask":
            M = rng.uniform(0.2, 2.0, (p, p))
            M = 0.5 * (M + M.T)
            lam = np.full(K, lam[0])
            args["lambda1_mask"], ref_mask = M, [M] * K
        elif kind == "maskK":
            M = rng.uniform(0.2, 2.0, (K, p, p))
            M = 0.5 * (M + M.transpose(0, 2, 1))
            args["lambda1_mask"], ref_mask = M, [M[k] for k in range(K)]
        elif kind == "dims":
            dims = rng.integers(2, p + 1, K)
            ref_S = [S[k][:dims[k], :dims[k]] for k in range(K)]
            S = pad_blocks(ref_S, p, True)
            args.update(dims=dims, Omega_0=pad_blocks([np.eye(d) for d in dims], p, True), X_0=np.zeros((K, p, p)))
        elif kind == "scaled":
            S = S.copy()
            S[1] *= 70.0
            ref_S = [S[k] for k in range(K)]
        res = quiet(ADMM_SGL_batch, S, lam, **args)
        for k in range(K):
            q = ref_S[k].shape[0]
            ref, rinfo = quiet(orc.ADMM_SGL, ref_S[k], float(lam[k]), np.eye(q),
                               lambda1_mask=None if ref_mask[k] is None else ref_mask[k][:q, :q], **kw)
            assert res[k][1]["status"] == rinfo["status"], (kind, p, K, k)
            assert np.abs(res[k][0]["Theta"] - ref["Theta"]).max() <= 1e-8 * max(1.0, np.abs(ref["Theta"]).max()), (kind, p, K, k)
