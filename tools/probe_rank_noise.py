"""Round 3: numpy.linalg.matrix_rank on the device's L (single_grid_search's RANK, helper/model_selection.py:638).
Below p = 128 the L-step reconstructs V diag((d - mu)_+) V^T from the Jacobi eigendecomposition (exact zeros up to
1e-16 |L|); above, L comes out of the sign iteration and its null space carries the iteration's residual.  Prints the
spectrum of L around numpy's tolerance p*eps*|L| for both L-steps and the ranks each rule gives."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import solver, synth


def run(p, mu1, lam, rank_eig, tol=1e-8):
    S, _ = synth.make_problem("GGL", 1, p, seed=3)
    solver.ENGINE_OPTIONS.clear()
    if rank_eig:
        solver.ENGINE_OPTIONS["rank_eig"] = 1.0
    sol, info = solver.ADMM_SGL(S[0], lam, np.eye(p), tol=tol, rtol=tol, latent=True, mu1=mu1, verbose=False)
    L = sol["L"]
    ev = np.abs(np.linalg.eigvalsh(L))
    ev.sort()
    tolnp = ev.max() * p * np.finfo(float).eps
    r_np = int(np.linalg.matrix_rank(L, hermitian=True))
    big = ev[ev > 1e-6 * ev.max()]
    small = ev[ev <= 1e-6 * ev.max()]
    print(f"p={p} mu1={mu1} rank_eig={int(rank_eig)} it={info.get('iterations', '?')} |L|2={ev.max():.3e} numpy tol={tolnp:.2e} "
          f"matrix_rank={r_np} #(>1e-6|L|)={big.size} smallest kept={big.min() if big.size else 0:.3e} "
          f"null-space max={small.max() if small.size else 0:.3e} ({(small.max() / ev.max()) if small.size and ev.max() > 0 else 0:.1e} |L|)")
    return r_np, big.size


if __name__ == "__main__":
    for p in (100, 200, 500):
        for mu1 in (0.5, 1.0, 2.0):
            for re in (True, False):
                run(p, mu1, 0.1, re)
