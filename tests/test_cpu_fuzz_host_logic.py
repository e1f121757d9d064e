"""The randomised cases of tests/fuzz_checks.py through the product's HOST logic on the CPU: gglasso_amd.solver's ADMM loop,
gglasso_amd.batch's per-point decisions / compaction bookkeeping / latent finalisation, block_SGL's bucketing, ext_solver's padding and
model_selection's tables and selections -- with the test-only OracleEngine (tests/oracle_engine.py) standing where the HIP engine is.
What the -m gpu suite checks with the kernels underneath (tests/test_gpu_fuzz.py), checked here for the Python around them.
Reference: solver/admm_solver.py:13-313, solver/single_admm_solver.py:15-475, solver/ext_admm_solver.py:18-323,
helper/model_selection.py:55-692."""
import pytest

import fuzz_checks


@pytest.fixture()
def host_logic(monkeypatch):
    from gglasso_amd import solver
    from oracle_engine import OracleEngine
    monkeypatch.setattr(solver, "ENGINE", OracleEngine)
    monkeypatch.setattr(fuzz_checks, "P", [q for q in fuzz_checks.P if q <= 47])     # (numpy eigh per instance and iteration)
    return solver


@pytest.mark.parametrize("kind,cases,seed", [("solver", 100, 301), ("batch", 40, 302), ("block", 120, 303), ("ext", 50, 304),
                                              ("grid", 50, 305), ("mgrid", 50, 306), ("kgrid", 40, 307), ("stats", 60, 308)])
def test_host_logic_on_random_cases(host_logic, kind, cases, seed):
    lines = []
    bad, notes, worst = fuzz_checks.run_cases(cases, seed, out=lines.append, kind=kind, big=False)
    assert bad == 0, "\n".join(lines)
    assert worst <= fuzz_checks.TOL
