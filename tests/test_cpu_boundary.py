"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/ggl_hip.h declares, the ctypes table matches the header, the product path refuses to run
without a GPU (no silent fallback), and the host loop reproduces the reference control flow when
driven by a test-only oracle engine."""
import contextlib
import io
import os
import re

import numpy as np
import pytest

from conftest import load_golden, ROOT
from oracle import ggl_oracle as orc


def _declared_symbols(dev=False):
    """Entry points include/ggl_hip.h declares; dev: the ones inside its #ifdef GGL_DEV section (libggl_hip_dev.so)."""
    txt = open(os.path.join(ROOT, "include", "ggl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    dev_txt = "".join(re.findall(r"#ifdef GGL_DEV(.*?)#endif", txt, flags=re.S))
    if dev:
        txt = dev_txt
    else:
        txt = re.sub(r"#ifdef GGL_DEV.*?#endif", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ggl_[A-Za-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from gglasso_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    import ctypes
    names = _declared_symbols()
    assert len(names) >= 25
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/ggl_hip.h but not exported"
    assert sorted(lib.EXPORTS) == names, "ctypes signature table out of sync with the header"
    assert lib.load().ggl_version() == lib.ABI_VERSION == 300
    # the development-only entry points are declared under GGL_DEV and are NOT in the shipped library
    dev = _declared_symbols(dev=True)
    assert sorted(lib._DEV_SIGNATURES) == dev and dev
    for n in dev:
        assert not hasattr(so, n), f"{n} is development scaffolding and must not be exported by libggl_hip.so"


def test_shipped_library_reads_no_environment_and_has_no_dev_kernels(lib):
    """ablation modes, probe kernels and GGL_* environment knobs live in the GGL_DEV build only."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    blob = open(lib.LIB_PATH, "rb").read()
    for needle in (b"k_mfma_f64_peak", b"k_mfma_valu_mix", b"GGL_SPEC_FACTOR", b"GGL_NS_MODE", b"k_symm_i8", b"k_slice_i8",
                   b"ggl_dev_symm_i8", b"ggl_dev_omega_i8"):
        assert needle not in blob, needle


def test_header_constants_match_python(lib):
    txt = open(os.path.join(ROOT, "include", "ggl_hip.h")).read()
    consts = dict(re.findall(r"#define\s+(GGL_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", txt))
    assert int(consts["GGL_REG_GGL"]) == lib.REG_GGL and int(consts["GGL_REG_FGL"]) == lib.REG_FGL
    assert int(consts["GGL_EIG_JACOBI"]) == lib.EIG_JACOBI and int(consts["GGL_EIG_ROCSOLVER"]) == lib.EIG_ROCSOLVER
    assert int(consts["GGL_JACOBI_MAX_P"]) == lib.JACOBI_MAX_P
    assert int(consts["GGL_NS_MIN_P"]) == lib.NS_MIN_P
    assert int(consts["GGL_BUF_GROUPSQ"]) == lib.BUF_GROUPSQ
    assert int(consts["GGL_NPHASE"]) == len(lib.PHASES)
    assert int(consts["GGL_E_ARG"]) == lib.E_ARG
    for name, idx in lib.OPTIONS.items():
        assert int(consts["GGL_OPT_" + name.upper()]) == idx
    assert lib.eig_flags(3, 2, 9) == 3 | (2 << 8) | (9 << 12)


def _has_gpu(lib):
    return lib.load().ggl_device_count() > 0


def test_product_path_fails_loudly_without_gpu(lib):
    if _has_gpu(lib):
        pytest.skip("GPU present")
    from gglasso_amd import solver, ops
    S = np.stack([np.eye(4)] * 2)
    with pytest.raises(RuntimeError, match="needs an AMD GPU"):
        solver.ADMM_MGL(S, 0.1, 0.1, 'GGL', S.copy())
    with pytest.raises(RuntimeError, match="needs an AMD GPU"):
        ops.prox_p(S, 0.1, 0.1, 'GGL')
    with pytest.raises(RuntimeError, match="needs an AMD GPU"):
        ops.eigh(np.eye(3))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "gglasso_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in src and "import oracle" not in src and "ggl_oracle" not in src, f


def test_argument_asserts_happen_before_any_device_work(lib):
    from gglasso_amd import solver
    S = np.stack([np.eye(4)] * 2)
    with pytest.raises(AssertionError):
        solver.ADMM_MGL(S, 0.1, 0.1, 'XXX', S.copy())
    with pytest.raises(AssertionError):
        solver.ADMM_MGL(S, 0.1, 0.0, 'GGL', S.copy())
    with pytest.raises(AssertionError):
        solver.ADMM_MGL(S, 0.1, 0.1, 'GGL', S[:1].copy())
    with pytest.raises(AssertionError):
        solver.ADMM_SGL(np.eye(4), -1.0, np.eye(4))
    with pytest.raises(AssertionError):
        solver.ADMM_SGL(np.eye(4), 0.1, np.eye(4), lambda1_mask=np.ones((3, 3)))
    with pytest.raises(AssertionError):
        solver.ADMM_SGL(np.eye(4), 0.1, np.eye(4), latent=True)


@pytest.fixture()
def oracle_engine(monkeypatch):
    from gglasso_amd import solver
    from oracle_engine import OracleEngine
    monkeypatch.setattr(solver, "ENGINE", OracleEngine)
    return solver


def _quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


@pytest.mark.parametrize("reg", ["GGL", "FGL"])
@pytest.mark.parametrize("latent", [False, True])
def test_host_loop_control_flow_matches_reference(oracle_engine, reg, latent):
    """Host loop (rho rule, dim*eps stopping rule, status strings, info arrays, final print) pinned by
    the reference trajectories G8/G9 -- the array work is done by the test-only oracle engine."""
    g = load_golden("g8_g9_admm_mgl")
    S, Om0 = g[f"S_{reg}"], g["Omega_0"]
    l1, l2, mu1 = g["params"]
    tag = f"{reg}_{'lat' if latent else 'nol'}"
    (sol, info), out = _quiet(oracle_engine.ADMM_MGL, S, l1, l2, reg, Om0, max_iter=10, tol=1e-20, rtol=1e-20,
                              latent=latent, mu1=float(mu1), measure=True)
    for nm in ('Omega', 'Theta', 'L', 'X'):
        assert np.abs(sol[nm] - g[f"{tag}_it10_{nm}"]).max() <= 1e-10
    assert np.allclose(info['residual'], g[f"{tag}_it10_residual"], rtol=1e-8)
    assert np.allclose(info['objective'], g[f"{tag}_it10_objective"], rtol=1e-10)
    assert sorted(info) == ['objective', 'residual', 'runtime', 'status']
    assert "ADMM terminated after 10 iterations with status: max iterations reached." in out
    (sol, info), out = _quiet(oracle_engine.ADMM_MGL, S, l1, l2, reg, Om0, tol=1e-10, rtol=1e-10, latent=latent,
                              mu1=float(mu1), verbose=True)
    assert info == {'status': str(g[f"{tag}_conv_status"])}
    assert np.linalg.norm(sol['Theta'] - g[f"{tag}_conv_Theta"]) <= 1e-8
    assert out.count("\n") >= int(g[f"{tag}_conv_iters"]) + 2     # header + one line per iteration + final


def test_host_loop_sgl_and_kkt(oracle_engine):
    g = load_golden("g10_admm_sgl")
    S, mask = g["S"], g["mask"]
    p = S.shape[0]
    (sol, info), _ = _quiet(oracle_engine.ADMM_SGL, S, 0.05, np.eye(p), max_iter=10, tol=1e-20, rtol=1e-20,
                            lambda1_mask=mask, measure=True)
    assert sorted(sol) == ['Omega', 'Theta', 'X']
    for nm in sol:
        assert np.abs(sol[nm] - g[f"mask_it10_{nm}"]).max() <= 1e-10
    assert sorted(info) == ['residual', 'runtime', 'status']
    g2 = load_golden("g8_g9_admm_mgl")
    l1, l2, _ = g2["params"]
    (sol, info), _ = _quiet(oracle_engine.ADMM_MGL, g2["S_GGL"], l1, l2, 'GGL', g2["Omega_0"], tol=1e-6,
                            stopping_criterion='kkt', measure=True)
    assert len(info['residual']) == int(g2["kkt_run_iters"])


def test_synth_generator_properties():
    from gglasso_amd import synth
    for reg in ("GGL", "FGL"):
        S, Th = synth.make_problem(reg, 4, 40, seed=5)
        S2, _ = synth.make_problem(reg, 4, 40, seed=5)
        assert np.array_equal(S, S2)
        assert np.array_equal(S, S.transpose(0, 2, 1))
        assert np.linalg.eigvalsh(S).min() > 0
        assert np.linalg.eigvalsh(Th).min() > 0
        assert (np.abs(Th) > 0).mean() < 0.2


def test_host_logic_sgl_batch_and_block_sgl(oracle_engine):
    """Per-instance rho / stopping bookkeeping of ADMM_SGL_batch and the component split + reassembly of
    block_SGL, pinned by the reference vectors G10/G11 (array work: test-only oracle engine)."""
    from gglasso_amd.batch import ADMM_SGL_batch
    g = load_golden("g10_admm_sgl")
    S = g["S"]
    p = S.shape[0]
    lams = np.array([0.05, 0.2, 0.01])
    res = ADMM_SGL_batch(S, lams, tol=1e-10, rtol=1e-10)
    for k, lam in enumerate(lams):
        ref, rinfo = orc.ADMM_SGL(S, lam, np.eye(p), tol=1e-10, rtol=1e-10)
        assert res[k][1]['iterations'] == rinfo['iterations'] and res[k][1]['status'] == 'optimal'
        assert np.abs(res[k][0]['Theta'] - ref['Theta']).max() <= 1e-10
    assert np.linalg.norm(res[0][0]['Theta'] - g["plain_conv_Theta"]) <= 1e-8
    g11 = load_golden("g11_block_sgl")
    S, lam = g11["S"], float(g11["lam"])
    (sol, out) = _quiet(oracle_engine.block_SGL, S, lam, np.eye(S.shape[0]), tol=1e-10, rtol=1e-10)
    for nm in ("Omega", "Theta", "X"):
        assert np.abs(sol[nm] - g11[nm]).max() <= 1e-9
    (solm, _) = _quiet(oracle_engine.block_SGL, S, lam, np.eye(S.shape[0]), tol=1e-10, rtol=1e-10,
                       lambda1_mask=g11["mask"])
    assert np.abs(solm["Theta"] - g11["mask_Theta"]).max() <= 1e-9


def _ns_schedule(lib, l, degrees):
    import ctypes
    deg = (ctypes.c_int * 24)()
    co = (ctypes.c_double * (24 * 6))()
    units = ctypes.c_int()
    n = lib.load().ggl_dev_ns_schedule(float(l), int(degrees), 24, deg, co, ctypes.byref(units))
    assert n > 0, (l, degrees, n)
    return list(deg)[:n], np.array(co[:6 * n]).reshape(n, 6), units.value


def test_newton_schulz_schedule_at_a_stopping_tolerance(lib):
    """GGL_OPT_NS_TOL (host logic, no GPU): a schedule planned for tolerance tol maps [l,1] into [1 - tol, 1] (to rounding),
    never costs more than the fp64-resolution schedule, and at the headline's l (0.70-0.72 from the carried bound vector)
    the default 2e-12 is what buys the seventh-instead-of-eighth product."""
    import ctypes
    def plan(l, tol):
        deg = (ctypes.c_int * 24)()
        co = (ctypes.c_double * (24 * 6))()
        units = ctypes.c_int()
        n = lib.load().ggl_dev_ns_schedule_tol(float(l), 9, float(tol), 24, deg, co, ctypes.byref(units))
        assert n > 0, (l, tol, n)
        return list(deg)[:n], np.array(co[:6 * n]).reshape(n, 6), units.value
    for l in np.concatenate([np.geomspace(1e-4, 0.99, 40), [0.70, 0.715, 0.72]]):
        exact = plan(l, 0.0)[2]
        assert exact == _ns_schedule(lib, l, 9)[2]
        last = exact
        for tol in (1e-14, 2e-12, 1e-9, 1e-6):
            deg, co, units = plan(l, tol)
            assert units <= last, (l, tol, units, last)
            last = units
            x = np.unique(np.concatenate([np.linspace(l, 1, 20001), np.geomspace(l, 1, 20001)])).astype(np.longdouble)
            for i in range(len(deg)):
                x2 = x * x
                x = x * (co[i, 0] + x2 * (co[i, 1] + x2 * (co[i, 2] + x2 * (co[i, 3] + x2 * co[i, 4]))))
            assert float(np.abs(1 - x).max()) <= tol * 1.001 + 2e-15, (l, tol, deg)
    assert plan(0.70, 0.0)[2] == 8 and plan(0.70, 2e-12)[2] == 7 and plan(0.72, 1e-12)[2] == 7
    assert plan(0.66, 2e-12)[2] == 8          # without the carried bound vector (l ~ 0.64-0.67) the default saves nothing


@pytest.mark.parametrize("degrees", [3, 5, 9])
def test_newton_schulz_schedule_converges_on_its_interval(lib, degrees):
    """Host logic of the eigendecomposition-free Omega-step (no GPU): composing the planned step polynomials
    x -> x t(x^2) maps every x in [l,1] to 1 within a few ulps, every step stays inside the interval it
    promises to the next one, and the product count is what the step costs add up to."""
    cost = {3: (0, 3, 2), 5: (1, 4, 3), 9: (2, 5, 4)}         # first / middle / last, on top of A', B'
    for l in np.concatenate([np.geomspace(1e-6, 0.999, 60), 1 - np.geomspace(1e-12, 1e-3, 10), [1.0]]):
        deg, co, units = _ns_schedule(lib, l, degrees)
        assert set(deg) <= {3, 5, 9} and max(deg) <= degrees
        n = len(deg)
        expect = 2 + sum(cost[d][0] if i == 0 else (cost[d][2] if i == n - 1 else cost[d][1]) for i, d in enumerate(deg))
        assert units == expect, (l, deg, units, expect)
        x = np.unique(np.concatenate([np.linspace(l, 1, 20001), np.geomspace(l, 1, 20001)])).astype(np.longdouble)
        for i in range(n):
            x2 = x * x
            x = x * (co[i, 0] + x2 * (co[i, 1] + x2 * (co[i, 2] + x2 * (co[i, 3] + x2 * co[i, 4]))))
            assert x.min() >= co[i, 5] - 1e-13 and x.max() <= 1 + 1e-13, (l, i, deg)
        assert float(np.abs(1 - x).max()) < 2e-15, (l, deg)


def test_newton_schulz_schedule_higher_degrees_never_cost_more(lib):
    for l in np.geomspace(1e-4, 0.99, 40):
        u3 = _ns_schedule(lib, l, 3)[2]
        u5 = _ns_schedule(lib, l, 5)[2]
        u9 = _ns_schedule(lib, l, 9)[2]
        assert u9 <= u5 <= u3, (l, u3, u5, u9)
    # the headline workload's interval: two degree-nine steps, eight products
    deg, _, units = _ns_schedule(lib, 0.67, 9)
    assert deg == [9, 9] and units == 8


def test_group_schedules_partition_rule(lib):
    """GGL_OPT_GROUP_SCHED (host logic, no GPU): a batch of independent problems whose instances need different product counts
    (the reference solves every grid point with its own eigh, helper/model_selection.py:619-633) is cut into contiguous groups
    with their own schedules -- only where the deterministic time model sum_g U_g (F + len_g I(p)) gains at least 6 %."""
    import ctypes
    L = lib.load()
    ip = ctypes.POINTER(ctypes.c_int)

    def part(u, p, mg=3):
        u = np.ascontiguousarray(u, dtype=np.int32)
        out = np.zeros(3, dtype=np.int32)
        g = L.ggl_dev_group_partition(u.ctypes.data_as(ip), len(u), int(p), int(mg), out.ctypes.data_as(ip))
        assert g >= 1 and out[:g].sum() == len(u) and np.all(out[:g] >= 1)
        return g, out[:g].tolist()

    def model(u, lens, p):
        I, F, t, k0 = 2.5e-14 * p ** 3, 6.5e-6, 0.0, 0
        for n in lens:
            t += max(u[k0:k0 + n]) * (F + n * I)
            k0 += n
        return t

    c2 = [7] * 7 + [8] * 3 + [9] * 4 + [10] * 3 + [11] * 3                 # a lambda1 grid ordered by conditioning
    g, lens = part(c2, 1000)
    assert g == 3 and model(c2, lens, 1000) <= 0.94 * model(c2, [20], 1000)
    # the optimum over all cuts into at most three runs (brute force)
    best = min(model(c2, [i, j - i, 20 - j], 1000) for i in range(1, 19) for j in range(i + 1, 20))
    assert abs(model(c2, lens, 1000) - best) <= 1e-12 * best
    assert part(c2, 1000, 2)[0] == 2 and part(c2, 1000, 1) == (1, [20])
    assert part(c2, 200) == (1, [20]) and part(c2, 500) == (1, [20])          # launches too small for a split to pay
    assert part([7] * 32, 500) == (1, [32]) and part([8] * 20, 1000) == (1, [20])   # nothing to gain
    assert part([7, 11], 1000) == (2, [1, 1])
    rng = np.random.default_rng(0)
    for _ in range(200):
        K, p = int(rng.integers(2, 48)), int(rng.choice([300, 700, 1000, 1500]))
        u = np.sort(rng.integers(6, 14, K)) if rng.random() < 0.7 else rng.integers(6, 14, K)
        g, lens = part(u, p)
        if g > 1:
            assert model(list(u), lens, p) <= 0.94 * model(list(u), [K], p) * (1 + 1e-12)
    # product counts of the schedules the groups are built from
    assert L.ggl_dev_ns_units(0.72, 9, 2e-12) == 7 and L.ggl_dev_ns_units(0.3, 9, 2e-12) == 11
    assert L.ggl_dev_ns_units(0.05, 9, 2e-12) == -1                            # condition number 400: the stable schedule's range
    last = 0
    for l in np.geomspace(0.99, 0.06, 60):
        u = L.ggl_dev_ns_units(float(l), 9, 2e-12)
        assert u >= last
        last = u


def test_batched_single_grid_search_matches_reference_tables(oracle_engine):
    """the whole (lambda1[, mu1]) grid as one batch reproduces the AIC / eBIC / sparsity / rank
    tables, the selected point and the selected Theta of the reference's sequential single_grid_search
    (fixture G12, generated from the real reference).  Array work: test-only oracle engine."""
    from grid_checks import check_single_grid_search, check_k_single_grid
    check_single_grid_search(load_golden)
    check_k_single_grid(load_golden)


def test_selection_criteria_definitions():
    from gglasso_amd import model_selection as ms
    rng = np.random.default_rng(3)
    p, N = 12, 80
    A = rng.standard_normal((p, 3 * p))
    S = A @ A.T / (3 * p)
    Th = np.linalg.inv(S + 0.5 * np.eye(p))
    Th[np.abs(Th) < 0.05] = 0.0
    Th = 0.5 * (Th + Th.T)
    E = (np.count_nonzero(Th) - p) / 2
    fit = N * np.sum(S * Th) - N * np.linalg.slogdet(Th)[1]
    assert np.isclose(ms.aic_single(S, Th, N), fit + E)
    assert np.isclose(ms.ebic_single(S, Th, N, 0.3), fit + E * (np.log(N) + 4 * np.log(p) * 0.3))
    mask = np.ones((p, p)); mask[:3] = mask[:, :3] = 0.25
    Em = (((Th != 0) * mask).sum() - np.trace((Th != 0) * mask)) / 2
    assert np.isclose(ms.ebic_single(S, Th, N, 0.5, lambda1_mask=mask), fit + Em * (np.log(N) + 4 * np.log(p) * 0.5))
    assert ms.robust_logdet(np.diag([1.0, 1e-13])) == -np.inf
    assert np.isclose(ms.sparsity(Th), 2 * E / (p * p - p))
    S3, T3 = np.stack([S, S]), np.stack([Th, Th])
    assert np.isclose(ms.aic(S3, T3, N), 2 * ms.aic_single(S, Th, N))
    assert np.isclose(ms.ebic(S3, T3, np.array([N, N]), 0.1), 2 * ms.ebic_single(S, Th, N, 0.1))


def test_exit_report_warning_branches():
    """The exit checks of the reference (solver/admm_solver.py:284-301, single_admm_solver.py:244-263): a warning
    per stack that is not symmetric to 1e-5, the 'not positive definite' print when min eig(Theta - L) <= 0 and the
    'not positive semidefinite' print when min eig(L) < -tol -- host logic over the five numbers of ggl_exit_checks."""
    import warnings
    from gglasso_amd import solver

    class Eng:
        def __init__(self, vals):
            self.vals = vals

        def exit_checks(self, latent):
            return np.array(self.vals, dtype=float)

    with warnings.catch_warnings():
        warnings.simplefilter("error")          # a clean state must not warn
        _, out = _quiet(solver._exit_report, Eng([0, 1e-6, 0, 0.3, 0.0]), True, 1e-5, False)
    assert out == ""
    with pytest.warns(UserWarning) as rec:
        _, out = _quiet(solver._exit_report, Eng([2e-5, 3e-5, 4e-5, -0.1, -1e-3]), True, 1e-5, True)
    msgs = [str(w.message) for w in rec]
    assert [m.split(" variable")[0] for m in msgs] == ["Omega", "Theta", "L"]
    assert "largest deviation is 3e-05" in msgs[1]
    assert "WARNING: Theta (Theta - L resp.) is not positive definite. Solve to higher accuracy! (min EV is -0.1)" in out
    assert "WARNING: L is not positive semidefinite. Solve to higher accuracy! (min EV is -0.001)" in out
    # MGL flavour: no eigenvalue in the text; L only checked when latent, and only beyond the tolerance
    _, out = _quiet(solver._exit_report, Eng([0, 0, 0, 0.0, -1e-6]), True, 1e-5, False)
    assert out == "WARNING: Theta (Theta - L resp.) is not positive definite. Solve to higher accuracy!\n"
    _, out = _quiet(solver._exit_report, Eng([0, 0, 0, 1.0, -1.0]), False, 1e-5, False)
    assert out == ""


def test_speculative_reject_codes_are_not_swallowed(monkeypatch):
    """a positive return code (GGL_SPEC_RETRY) from ggl_step_finish / ggl_admm_step must never be taken
    for an accepted step.  step_finish returns None for it (the sharded loop repeats the iteration and raises if the
    repeat is rejected as well); step raises."""
    from gglasso_amd import solver

    class Lib:
        rc = 1

        def ggl_step_finish(self, *a):
            return self.rc

        def ggl_admm_step(self, *a):
            return self.rc

    eng = object.__new__(solver.HipEngine)
    eng.lib, eng.h, eng._norms = Lib(), None, np.arange(5.0)
    eng._norms_p, eng._ptr_cache = None, {}
    assert eng.step_finish(1.0, 0.1, 0.1, 'GGL', False, None, 1) is None
    with pytest.raises(RuntimeError, match="rejected twice"):
        eng.step(1.0, 0.1, 0.1, 'GGL', False, None, None)
    Lib.rc = 0
    assert np.array_equal(eng.step_finish(1.0, 0.1, 0.1, 'GGL', False, None, 1), np.arange(5.0))
    eng.h = None      # nothing to destroy


@pytest.mark.parametrize("latent", [False, True])
def test_ext_host_loop_and_padding_match_reference(oracle_engine, latent):
    """gglasso_amd.ext_solver.ext_ADMM_MGL (padding of the non-conforming instances into one stack, host loop, status
    strings, per-instance exit messages, un-padding) over the test-only oracle engine, pinned by the reference's
    trajectories G14 and the conforming case G15."""
    import ext_checks
    from gglasso_amd import ext_solver
    ext_checks.check_g14(load_golden, ext_solver.ext_ADMM_MGL, latent)
    ext_checks.check_g14_kkt(load_golden, ext_solver.ext_ADMM_MGL, latent)
    ext_checks.check_g15(load_golden, ext_solver.ext_ADMM_MGL, latent)


@pytest.mark.parametrize("latent", [False, True])
def test_ext_batched_grid_host_logic(oracle_engine, latent):
    """ext_ADMM_MGL_batch / grid_search(solver=ext_ADMM_MGL, dict S): grid points as slabs of one padded stack, per-point
    stopping, un-padding, selection equal to the sequential walk (host logic over the test-only oracle engine)."""
    import ext_checks
    from gglasso_amd import ext_solver, model_selection
    ext_checks.check_ext_batch(load_golden, ext_solver.ext_ADMM_MGL, ext_solver.ext_ADMM_MGL_batch,
                               None if latent else model_selection.grid_search, latent)


def test_ext_asserts_like_reference(oracle_engine):
    from gglasso_amd import ext_solver
    g = load_golden("g14_ext_admm_nonconforming")
    import ext_checks
    K, p, S, G, Om0 = ext_checks.g14_inputs(g)
    with pytest.raises(AssertionError):
        ext_solver.ext_ADMM_MGL(S, 0.1, 0.1, 'FGL', Om0, G)
    with pytest.raises(AssertionError):
        ext_solver.ext_ADMM_MGL(S, 0.1, -0.1, 'GGL', Om0, G)
    with pytest.raises(AssertionError):
        ext_solver.ext_ADMM_MGL(S, 0.1, 0.1, 'GGL', Om0, G, latent=True)
    Gbad = G.copy()
    Gbad[0, 0, :] = Gbad[1, 0, :]
    with pytest.raises(AssertionError, match="diagonal"):
        ext_solver.ext_ADMM_MGL(S, 0.1, 0.1, 'GGL', Om0, Gbad)
    with pytest.raises(AssertionError, match="integer"):
        ext_solver.ext_ADMM_MGL(S, 0.1, 0.1, 'GGL', Om0, G.astype(float))


def test_batched_mgl_grid_search_matches_reference_tables(oracle_engine):
    """gglasso_amd.model_selection.grid_search (lambda1 x lambda2 grid as one batch, per-point rho / stopping, criteria,
    selection, thresholding, sequential mode) over the test-only oracle engine against the reference's tables G16."""
    from grid_checks import check_mgl_grid_search
    _quiet(check_mgl_grid_search, load_golden)


def test_bench_bare_multi_gpu_start_launches_its_own_ranks():
    """``python bench.py --gpus N`` without a launcher must start its N ranks itself -- as a CHILD
    ``torch.distributed.run`` on 127.0.0.1, from a process that has not touched the GPU (no torch, no HIP library
    loaded) -- relay the child's output and return its exit code."""
    import subprocess
    import sys
    code = r"""
import sys, types
sys.argv = ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"]
import bench
seen = {}
def runner(cmd, env=None, stdout=None, text=None):
    seen["cmd"], seen["env"] = cmd, env
    return types.SimpleNamespace(stdout='{"metric": "x", "n_gpus": 4}\n', returncode=7)
rc = bench.spawn_ranks(4, sys.argv[1:], runner=runner)
assert rc == 7, rc
cmd = seen["cmd"]
assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"], cmd
assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1", cmd
assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py"), cmd
assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
assert "torch" not in sys.modules and "gglasso_amd" not in sys.modules, "the launcher process must stay off the GPU"
print("ok")
"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    assert lines[-1] == "ok" and lines[0].startswith('{"metric"'), out.stdout     # the child's JSON line is relayed


def test_block_sgl_ragged_components_share_padded_batches(oracle_engine):
    """block_SGL (single_admm_solver.py:326-475) on a covariance whose graph |S| > lambda1 splits into components of
    MANY different sizes, with and without a lambda1_mask: every component rides in an identity-padded slot of a batch
    of its size class (gglasso_amd.batch.ADMM_SGL_batch(dims=...)), with residuals, dim and stopping decision over its own
    block -- the result must be the oracle's component-by-component solve, and the number of engines the number of size
    classes, not of sizes (host logic + padding; array work: test-only oracle engine)."""
    from scipy.linalg import block_diag
    from gglasso_amd import solver, synth
    rng = np.random.default_rng(12)
    sizes = [2, 3, 3, 5, 7, 8, 11, 14, 1, 1, 20]
    blocks = []
    for q in sizes:
        A = rng.standard_normal((q, 3 * q))
        C = A @ A.T / (3 * q) + 0.5 * np.eye(q)
        blocks.append(C)
    S = block_diag(*blocks)
    perm = rng.permutation(S.shape[0])
    S = S[np.ix_(perm, perm)]
    p = S.shape[0]
    lam = 0.02
    made = []

    class Counting(solver.ENGINE):
        def __init__(self, S_, *a, **k):
            super().__init__(S_, *a, **k)
            made.append(tuple(S_.shape))
    solver.ENGINE, keep = Counting, solver.ENGINE
    try:
        (sol, out) = _quiet(oracle_engine.block_SGL, S, lam, np.eye(p), tol=1e-10, rtol=1e-10)
        ref = orc.block_SGL(S, lam, np.eye(p), tol=1e-10, rtol=1e-10)
        for nm in ("Omega", "Theta", "X"):
            assert np.abs(sol[nm] - ref[nm]).max() <= 1e-10, nm
        # 9 non-singleton components of 8 distinct sizes in the classes <=4, <=6, <=9, <=13, <=19, <=28: six batches
        assert len(made) == len({solver.block_bucket(q) for q in sizes if q > 1}) == 6, made
        assert out.count("ADMM terminated after") == sum(q > 1 for q in sizes)
        mask = rng.uniform(0.5, 1.5, (p, p))
        mask = 0.5 * (mask + mask.T)
        (solm, _) = _quiet(oracle_engine.block_SGL, S, lam, np.eye(p), tol=1e-10, rtol=1e-10, lambda1_mask=mask)
        refm = orc.block_SGL(S, lam, np.eye(p), tol=1e-10, rtol=1e-10, lambda1_mask=mask)
        for nm in ("Omega", "Theta", "X"):
            assert np.abs(solm[nm] - refm[nm]).max() <= 1e-10, nm
    finally:
        solver.ENGINE = keep


def test_host_period_of_broadcast_views():
    """solver._host_period: which host arrays travel once and are replicated on the device (ggl_set_S_ex / ggl_set_state_ex)."""
    from gglasso_amd.solver import _host_period
    p = 6
    A = np.arange(p * p, dtype=np.float64).reshape(p, p)
    h, per = _host_period(np.broadcast_to(A, (5, p, p)))
    assert per == 1 and h.shape == (p, p) and h.flags.c_contiguous and np.array_equal(h, A)
    full = np.ascontiguousarray(np.broadcast_to(A, (5, p, p)))
    h, per = _host_period(full)
    assert per == 0 and h.shape == (5, p, p)
    h, per = _host_period(np.broadcast_to(A, (1, p, p)))            # K = 1: nothing to replicate
    assert per == 0 and h.shape == (1, p, p)
    B = np.stack([A, A + 1, A + 2])
    h, per = _host_period(np.broadcast_to(B, (4, 3, p, p)))          # one problem's stacks for 4 grid points
    assert per == 3 and h.shape == (3, p, p) and np.array_equal(h, B)
    h, per = _host_period(np.ascontiguousarray(np.broadcast_to(B, (4, 3, p, p))))
    assert per == 0 and h.shape == (12, p, p)
    h, per = _host_period(np.asfortranarray(full))                   # any other layout is materialised
    assert per == 0 and h.flags.c_contiguous


def test_latent_rank_rule_and_threshold_choice():
    """solver.latent_rank (the RANK tables' rule: numpy.linalg.matrix_rank's, every returned L being an eigendecomposition's
    since round 4; the 1e-9 cut of rounds 1-3 on request) and model_selection._pick_threshold (tune_threshold's
    choice, helper/model_selection.py:718-735, from the device table) on the host."""
    from gglasso_amd import solver, model_selection as ms
    rng = np.random.default_rng(4)
    p = 40
    Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
    ev = np.zeros(p)
    ev[:7] = rng.uniform(0.1, 2.0, 7)
    L = (Q * ev) @ Q.T
    L = 0.5 * (L + L.T)
    assert solver.latent_rank(L) == 7 == np.linalg.matrix_rank(L, hermitian=True)
    noisy = L + 1e-13 * (Q[:, 7:9] @ Q[:, 7:9].T)                    # a null space at 1e-13 |L|, as the sign iteration leaves it
    assert np.linalg.matrix_rank(noisy, hermitian=True) > 7 and solver.latent_rank(noisy, solver.RANK_REL_TOL) == 7
    assert solver.latent_rank(noisy) == np.linalg.matrix_rank(noisy, hermitian=True)        # default: numpy's rule
    assert solver.latent_rank(np.zeros((p, p))) == 0
    # threshold choice: scores from a table {<S,T>, log det T, nnz, lambda_min} equal the host criteria; -inf log det -> nan
    S = np.cov(rng.standard_normal((p, 4 * p)))
    Th = np.linalg.inv(S + 0.5 * np.eye(p))
    Th[np.abs(Th) < 0.02] *= 1e-3
    taus = ms.default_tau_range()
    N = 3 * p
    tab = np.zeros((len(taus), 4))
    for j, tau in enumerate(taus):
        T = ms.thresholding(Th, tau)
        d = np.linalg.eigvalsh(T)
        tab[j] = [np.sum(S * T), -np.inf if d.min() <= 1e-12 else np.linalg.slogdet(T)[1], np.count_nonzero(T), d.min()]
    for method in ("eBIC", "AIC"):
        _, tau, scores = ms.tune_threshold(Th, S, N, method=method, gamma=0.3)
        assert taus[ms._pick_threshold(tab.copy(), N, p, method, 0.3)] == tau
