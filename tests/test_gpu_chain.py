"""k_omega_chain (csrc/gemm_sym.hip): the Omega-step's whole product chain as ONE persistent launch with per-instance
dependencies.  It runs the same tile body on the same operands with the same coefficients as the launch-per-product
path, so the iterates must agree BIT FOR BIT -- any stale read across a hand-off (a tile consumed before its producer's
stores were visible) would show up as a difference.  Shapes: full XCD rounds (K = 8, 16, 32), a ragged batch (K = 11: XCDs
with one and with two instances), p = 500 and a p that is not a multiple of the tile (p = 330); schedules: the default
(quintic + degree nine), fp64 resolution (two degree-nine steps with a pair product between them), quintic-only (three
steps, two pair products).  Against the oracle as well (solver/admm_solver.py:172-246).
"""
import contextlib
import io

import numpy as np
import pytest

from oracle import ggl_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("dev_library")]      # GGL_OPT_CHAIN: development library only


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _solve(opts, S, Om0, **kw):
    from gglasso_amd import solver
    stats = []
    real_close = solver.HipEngine.close

    def closing(self):
        if getattr(self, "h", None):
            stats.append(self.ns_stats())
        real_close(self)

    solver.HipEngine.close = closing
    old = dict(solver.ENGINE_OPTIONS)
    solver.ENGINE_OPTIONS.clear()
    solver.ENGINE_OPTIONS.update(opts)
    try:
        out, info = quiet(solver.ADMM_MGL, S, 0.05, 0.01, "GGL", Om0, **kw)
    finally:
        solver.ENGINE_OPTIONS.clear()
        solver.ENGINE_OPTIONS.update(old)
        solver.HipEngine.close = real_close
    return out, info, stats[-1]


@pytest.mark.parametrize("K,p,extra", [(32, 500, {}), (16, 500, {}), (8, 500, {}), (11, 330, {}), (16, 330, {"ns_tol": 0.0}),
                                       (8, 500, {"ns_degrees": 5}), (32, 500, {"ns_tol": 0.0})])
def test_chain_equals_launch_per_product_bitwise(K, p, extra):
    from gglasso_amd import synth
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=1239)
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    # fixed rho: speculative steps (the ones the chain serves) from the second iteration on; from the identity start one of
    # the first is usually rejected (the spectrum still grows by more than the 2 % margin) and four iterations without
    # speculation follow, hence 14 iterations; pipelining on
    kw = dict(max_iter=14, tol=1e-20, rtol=1e-20, update_rho=False, rho=2.0)
    # the launch-per-product reference runs the batch as ONE launch sequence with ONE schedule ("parts": 1; two concurrent
    # parts plan their schedules from their own instances' bounds: rounding-level differences) on the tile kernel the chain
    # is built on ("symm_variant": 17; the 32x32-tile kernel small batches would take rounds its epilogue differently: 1 ulp)
    # ("early_part": 0 -- the chain is one launch and has no first part to queue early; with it the launch path would plan
    # from bounds one iteration older and differ in the last digits)
    same = {"parts": 1, "symm_variant": 17, "early_part": 0}
    a, _, sa = _solve({"chain": 2, **same, **extra}, S, Om0, **kw)
    b, _, sb = _solve({"chain": 0, **same, **extra}, S, Om0, **kw)
    if extra.get("ns_degrees") != 5:      # (a quintic-only schedule may open with a cubic step: then the chain stands back)
        assert sa["last_variant"] == 40 and sa["last_parts"] == 1, sa    # the chain really ran
    assert sb["last_variant"] != 40, sb
    assert sa["spec_calls"] >= 4, sa
    for nm in ("Omega", "Theta", "X"):
        assert np.array_equal(a[nm], b[nm]), (nm, float(np.abs(a[nm] - b[nm]).max()))
    assert np.array_equal(a["Omega"], a["Omega"].transpose(0, 2, 1))


def test_chain_with_rho_updates_against_the_oracle():
    """The headline's own first iterations (rho rule on: speculative and non-speculative steps alternate, pre-launched
    chains get dropped when rho changes) on the chain, against the oracle."""
    from gglasso_amd import synth
    K, p = 32, 500
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=1239)
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    kw = dict(max_iter=14, tol=1e-20, rtol=1e-20)
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=16)
    except Exception:  # noqa: BLE001
        lim = contextlib.nullcontext()
    with lim:
        ref, _ = orc.ADMM_MGL(S, 0.05, 0.01, "GGL", Om0, **kw)
    out, _, st = _solve({"chain": 1}, S, Om0, **kw)
    assert st["last_variant"] == 40, st
    for nm in ("Omega", "Theta", "X"):
        err = float(np.abs(out[nm] - ref[nm]).max())
        assert err <= 1e-9 * max(1.0, float(np.abs(ref[nm]).max())), (nm, err)


def test_chain_forced_speculation_miss_repeats_the_step():
    """Deflated bounds (spec_factor 0.9): the chain's bound check rejects every speculative step, the Theta-step kernels
    skip, the host repeats the step bounds-first -- same iterates as without speculation."""
    from gglasso_amd import synth
    K, p = 16, 330
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=5)
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    kw = dict(max_iter=8, tol=1e-20, rtol=1e-20, update_rho=False, rho=1.5)
    a, _, sa = _solve({"chain": 2, "spec_factor": 0.9}, S, Om0, **kw)
    b, _, sb = _solve({"chain": 0, "speculate": 0}, S, Om0, **kw)
    assert sa["spec_misses"] >= 1, sa
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(a[nm] - b[nm]).max() <= 1e-11, nm
