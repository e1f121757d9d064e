"""GGL_OPT_GROUP_SCHED: a batch of independent problems whose instances differ in conditioning runs as contiguous groups with
their own Newton-Schulz schedules.  Reference: the grid walks solve every point with its own eigh
(helper/model_selection.py:619-633, solver/single_admm_solver.py:157-214) -- no point pays for another point's spectrum."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(K, p, seed):
    from gglasso_amd import synth
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=seed)
    return S


def _run(S, rhos, lams, iters, options, idx=None, start=None):
    """`iters` batched SGL iterations of the instances idx (default all) from the grid start Omega_0 = X_0 = I
    (helper/model_selection.py:595-596) or from the state ``start``; returns the state, per-iteration sums and the engine's
    statistics."""
    from gglasso_amd import solver
    idx = np.arange(len(S)) if idx is None else np.asarray(idx)
    K, p = len(idx), S.shape[-1]
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    Om0, Th0, X0 = (eye, eye, eye.copy()) if start is None else (np.ascontiguousarray(start[nm][idx]) for nm in ("Omega", "Theta", "X"))
    eng = solver.HipEngine(np.ascontiguousarray(S[idx]), Om0, Th0, X0, options=options)
    try:
        sums = [eng.sgl_batch_step(rhos[idx], lams[idx], False, None).copy() for _ in range(iters)]
        st = eng.state()
        return st, np.array(sums), eng.group_stats(), eng.ns_stats(), eng.spectral_bounds()
    finally:
        eng.close()


def test_grouped_schedules_are_the_sub_batches_own_schedules():
    """Eight single problems at p = 256 in three classes of conditioning (rho = 0.5, 1.5, 16 on copies of one covariance matrix:
    the instances of a class are identical), iterated to near their fixed points so that every instance's product count stays
    what it is: with grouping forced (13) the batch then runs as contiguous groups, the SAME split in every step; every
    instance's iterate is BITWISE the one it has in a batch that holds its group alone under one schedule, within the
    Omega-step's tolerance of the single-schedule batch, and within 1e-9 of the oracle."""
    from oracle import ggl_oracle as orc
    K, p, warm, iters = 8, 256, 30, 5
    S = np.repeat(_problem(1, p, 41), K, axis=0)
    rhos = np.array([0.5, 0.5, 0.5, 1.5, 1.5, 16.0, 16.0, 16.0])
    lams = np.full(K, 0.08)
    base = {"symm_variant": 17.0}
    st_w, _, _, _, cb = _run(S, rhos, lams, warm, {**base, "group_sched": 0.0})
    st_g, sums_g, gs, _, _ = _run(S, rhos, lams, iters, {**base, "group_sched": 13.0}, start=st_w)
    assert gs['steps'] == iters and gs['groups'] >= 2 and gs['changes'] == 0, gs
    assert sum(gs['len']) == K and len(set(gs['units'])) == len(gs['units']), gs      # different schedules, or no group
    cuts = np.cumsum(gs['len'])[:-1].tolist()
    assert set(cuts) <= {3, 5}, gs                                              # the groups are whole classes
    st_1, sums_1, gs1, _, _ = _run(S, rhos, lams, iters, {**base, "group_sched": 0.0}, start=st_w)
    assert gs1['steps'] == 0
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(st_g[nm] - st_1[nm]).max() <= 1e-10, nm
    # every group alone, one schedule: the same bits
    k0 = 0
    for n in gs['len']:
        idx = np.arange(k0, k0 + n)
        st_s, sums_s, gss, _, _ = _run(S, rhos, lams, iters, {**base, "group_sched": 0.0}, idx, start=st_w)
        for nm in ("Omega", "Theta", "X"):
            assert np.array_equal(st_g[nm][idx], st_s[nm]), (nm, k0, n)
        assert np.array_equal(sums_g[:, idx], sums_s)
        k0 += n
    # and the reference's iteration (single_admm_solver.py:157-214), point by point, over all warm + iters iterations
    for k in (0, 3, 7):
        ref, _ = orc.ADMM_SGL(S[k], lams[k], np.eye(p), X_0=np.eye(p), rho=rhos[k], max_iter=warm + iters, tol=1e-20, rtol=1e-20,
                              update_rho=False)
        for nm in ("Omega", "Theta", "X"):
            assert np.abs(st_g[nm][k] - ref[nm]).max() <= 1e-9, (k, nm)


def test_grouping_leaves_uniform_batches_and_small_launches_alone():
    """The rule itself (GGL_OPT_GROUP_SCHED = 1): instances of equal product count stay one launch sequence, and so does a
    heterogeneous batch whose launches are too small for a split to pay (p = 256); the library reports what it did."""
    K, p = 8, 256
    S = _problem(K, p, 43)
    lams = np.full(K, 0.08)
    _, _, gs, _, _ = _run(S, np.ones(K), lams, 4, {})
    assert gs['steps'] == 0 and gs['groups'] == 1
    _, _, gs, _, cb = _run(S, np.array([0.5, 0.7, 1.0, 1.5, 4.0, 8.0, 16.0, 24.0]), lams, 4, {})
    assert gs['steps'] == 0, gs
    c, beta = cb
    assert np.all(c >= 4 * beta) and len(np.unique(np.round(c / (4 * beta)))) >= 3          # (the batch WAS heterogeneous)


def test_grouped_mgl_step_with_early_part_is_consistent():
    """ggl_admm_step with heterogeneous instances (an FGL-less GGL problem whose instances are scaled differently): the early
    first part of the next chain and the rest that follows it use the SAME split, and the iterates agree with the ungrouped
    run to the Omega-step's tolerance and with the oracle to 1e-9."""
    from gglasso_amd import solver
    from oracle import ggl_oracle as orc
    K, p, iters = 6, 256, 12
    S = _problem(K, p, 47) * np.array([0.2, 0.4, 0.7, 1.0, 1.6, 2.5])[:, None, None]
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    out = {}
    for name, opt in (("grouped", 13.0), ("whole", 0.0)):
        eng = solver.HipEngine(S, eye, eye, np.zeros_like(S), options={"group_sched": opt})
        try:
            for _ in range(iters):
                eng.step(1.0, 0.05, 0.02, "GGL", False, None, np.ones(K))
            out[name] = (eng.state(), eng.group_stats(), eng.pipeline_stats())
        finally:
            eng.close()
    assert out["grouped"][1]['steps'] >= 3, out["grouped"][1]
    assert out["whole"][1]['steps'] == 0
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(out["grouped"][0][nm] - out["whole"][0][nm]).max() <= 1e-10, nm
    ref, _ = orc.ADMM_MGL(S, 0.05, 0.02, "GGL", eye, max_iter=iters, tol=1e-20, rtol=1e-20, update_rho=False)
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(out["grouped"][0][nm] - ref[nm]).max() <= 1e-9, nm


def test_product_library_refuses_the_development_only_options():
    """Round 6: the measured-and-rejected alternatives are options of libggl_hip_dev.so only; the product
    library accepts their default (0) and refuses anything else with GGL_E_ARG (an AssertionError in the Python binding), and
    reports 0 for them."""
    from gglasso_amd import solver
    K, p = 2, 40
    S = _problem(K, p, 51)
    eye = np.repeat(np.eye(p)[None], K, axis=0)
    eng = solver.HipEngine(S, eye, eye, np.zeros_like(S))
    try:
        for name in ("chain", "fused_cw", "part_priority", "rank_cw", "bound_side", "parts_bias", "parts_order"):
            eng.set_option(name, 0)
            with pytest.raises(AssertionError, match="development"):
                eng.set_option(name, 1)
            assert eng.get_option(name) == 0.0
        eng.set_option("group_sched", 2)
        assert eng.get_option("group_sched") == 2.0
        with pytest.raises(AssertionError):
            eng.set_option("group_sched", 7)
    finally:
        eng.close()
