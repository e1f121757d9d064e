"""A quick pass of the randomised parity cases (tests/fuzz_checks.py; tools/fuzz_parity.py runs as many as asked for -- round 6:
1 700 cases over six streams, profiles/r6_fuzz_parity.txt): ADMM_MGL / ADMM_SGL through the C ABI against the oracle at shapes
around every tile / pair / wave boundary (p = 1 .. 130), K = 1 .. 9, both penalties, latent on / off, masks, singular S, rho
updates.  Reference: solver/admm_solver.py:13-313, solver/single_admm_solver.py:15-275."""
import pytest

import fuzz_checks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_and_parameters_against_the_oracle(seed):
    lines = []
    bad, notes, worst = fuzz_checks.run_cases(40, seed, out=lines.append, big=False)
    assert bad == 0, "\n".join(lines)
    assert worst <= fuzz_checks.TOL


def test_random_grids_through_the_batch_drivers_against_the_oracle():
    """ADMM_SGL_batch / ADMM_MGL_batch on random grids (G = 1 .. 12 points of random lambda / mu, per-point rho and stopping,
    iteration limits that cut most points off, compaction on / off) against the oracle's independent solve of every point."""
    lines = []
    bad, notes, worst = fuzz_checks.run_cases(24, 31, out=lines.append, kind="batch")
    assert bad == 0, "\n".join(lines)
    assert worst <= fuzz_checks.TOL


@pytest.mark.parametrize("kind,cases,seed", [("block", 60, 43), ("ext", 40, 53), ("ops", 60, 63), ("stats", 40, 73), ("grid", 25, 83),
                                              ("isolate", 40, 93), ("mgrid", 40, 103), ("kgrid", 25, 113), ("egrid", 12, 123)])
def test_random_block_and_nonconforming_problems_against_the_oracle(kind, cases, seed):
    """block_SGL on covariance matrices with planted components of very different size (singletons to 70, all solved together
    on the GPU; solver/single_admm_solver.py:326-475) and ext_ADMM_MGL on K = 2 .. 5 instances of different dimension with a
    random group structure (solver/ext_admm_solver.py:18-323), each against the oracle; the operators on their own (phiplus /
    prox_rank_norm from the matrix on stacks with engineered spectra -- seven decades, exact repeats, zeros, one sign, instances of
    very different conditioning in one stack -- and prox_p with ties and exact zeros across K; solver/ggl_helper.py); the device's selection statistics
    (<S,Theta>, log det, count_nonzero, lambda_min, rank of L, the thresholded tables; helper/model_selection.py:619-660, 698-737)
    against numpy on the solutions a random latent / non-latent grid returns; single_grid_search and grid_search
    (helper/model_selection.py:505-692, 55-298; the latter also with ext_ADMM_MGL on instances of different dimension) and
    K_single_grid (:300-503) on random grids against the host tables of the oracle's point-by-point solves;
    grids with poisoned points (NaN / Inf in their S): those end as 'solver error', the others are the grid's without them."""
    lines = []
    bad, notes, worst = fuzz_checks.run_cases(cases, seed, out=lines.append, kind=kind)
    assert bad == 0, "\n".join(lines)
    assert worst <= fuzz_checks.TOL
