#!/usr/bin/env python3
"""VERDICT r3 item 4 -- L-step: deflate the unresolved subspace instead of iterating it down.  CPU prototype (NumPy) on the
iterates of the C4 problem (FGL, p = 500, latent, mu1 = 0.5; K from argv, the instances are independent in the L-step).

L = (C - mu I)_+ = B (I + sign(B)) / 2, B = C - mu I.  The library's sign iteration X <- X t(X^2) resolves eigenvalues of B down
to l0 |B|; the schedule's length is set by the eigenvalues NEXT to the threshold (22 products at l0 = 1e-3, 27 at 1e-4, 31 at
1e-5, 36 at 1e-6).  After a COARSE pass every eigenvalue farther than l0 |B| from the threshold sits at +-1 and
R = I - X^2 is numerically of rank r = the handful of eigenvalues within l0 |B|:
    range finder   Y = R G, G p x q Gaussian;  Q1 = orth(Y), V = orth(R Q1) (absolute thresholds 1e-11 / 1e-10 on the remainders)
    exact small    H = V^T B V  (r x r), sign(H) by eigendecomposition
    correction     sign(B) = X + V (sign(H) - V^T X V) V^T     =>     L = B (I + X) / 2 + (B V) (sign(H) - V^T X V) V^T / 2
p^2 q work, no product.  Guarded by the trace check (trace sign(B) must be an integer) and by a probe of the range
(|R g - V V^T R g| for fresh g): rank(R) > q falls back to the iteration.

Reported per ADMM iteration: products of the coarse pass, the number of unresolved eigenvalues (max over instances), the
error of L against eigh with and without deflation, and the products the two-tier iteration needs for the same accuracy.

    python tools/proto_deflate.py [K] [p] [l0_coarse]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, _lib
from oracle import ggl_oracle as orc


def sign_plan(l0):
    lib = _lib.load()
    deg = (ctypes.c_int * 48)()
    co = (ctypes.c_double * (48 * 6))()
    units = ctypes.c_int()
    n = lib.ggl_dev_ns_schedule(float(l0), 109, 48, deg, co, ctypes.byref(units))       # degrees >= 100: the sign schedule
    assert n > 0
    return list(deg)[:n], np.array(co[:6 * n]).reshape(n, 6), units.value


def sym(M):
    return 0.5 * (M + M.T)


def sign_iterate(B, c, co):
    X = B / c
    for t in co:
        M = sym(X @ X)
        T = t[0] * np.eye(len(B)) + M @ (t[1] * np.eye(len(B)) + M @ (t[2] * np.eye(len(B)) + M @ (t[3] * np.eye(len(B)) + t[4] * M)))
        X = sym(X @ sym(T))
    return X


def _orth(Y, tau):
    """Gram-Schmidt (twice) with an ABSOLUTE acceptance threshold on the remainder of every column, as csrc/deflate.hip."""
    V = []
    for a in range(Y.shape[1]):
        y = Y[:, a].copy()
        for _ in range(2):
            for v in V:
                y -= (v @ y) * v
        n = np.linalg.norm(y)
        if n > tau:
            V.append(y / n)
    return np.stack(V, axis=1) if V else np.zeros((Y.shape[0], 0))


def deflate(B, X, q, rng, tau1=1e-11, tau2=1e-10):
    """The device algorithm (csrc/deflate.hip).  The FIRST version of this prototype took the numerical rank of Y = R G with a
    tolerance RELATIVE to its largest column (1e-10): whenever the largest residual eigenvalue was itself small (3e-8, 5e-10)
    that let pure noise columns into the basis -- directions that are not invariant under B -- and the 'correction' was wrong by
    1e-2.  R's resolved part sits at ~5e-15 ABSOLUTELY (X is scaled to |X| <= 1), so the thresholds are absolute; and a
    direction with a small residual r_i comes out of one application of R contaminated at the relative level 1e-13 / r_i, which
    one more application (V = orth(R Q1)) removes."""
    p = len(B)
    G = rng.standard_normal((p, q + 2))
    R = lambda Z: Z - X @ (X @ Z)              # R Z without forming R: two tall-skinny products
    Q1 = _orth(R(G[:, :q]), tau1)
    V = _orth(R(Q1), tau2) if Q1.shape[1] else Q1
    r = V.shape[1]
    probe = R(G[:, q:])
    for _ in range(2):
        probe = probe - V @ (V.T @ probe)
    leak = np.linalg.norm(probe, axis=0).max()
    if r == 0:
        return 0.5 * sym(B @ (np.eye(p) + X)), 0, leak, np.trace(X)
    BV = B @ V
    H = sym(V.T @ BV)
    w, U = np.linalg.eigh(H)
    sH = (U * np.sign(w)) @ U.T
    D = sH - sym(V.T @ (X @ V))
    L = 0.5 * sym(B @ (np.eye(p) + X)) + 0.5 * sym(BV @ D @ V.T)
    tr = np.trace(X) + np.trace(D)
    return L, r, leak, tr


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    l0c = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
    q = 6
    rng = np.random.default_rng(3)
    S, _ = synth.make_problem("FGL", K, p, N=2 * p, seed=1237)
    Om = np.stack([np.eye(p)] * K); Th = Om.copy(); X = np.zeros_like(S); L = np.zeros_like(S)
    rho, mu1 = 1.0, 0.5
    _, co_c, units_c = sign_plan(l0c)
    fine = {l0: sign_plan(l0)[2] for l0 in (1e-4, 1e-5, 1e-6)}
    print(f"FGL K={K} p={p} latent: coarse pass at l0 = {l0c:g}: {units_c} products; the iteration alone: " +
          ", ".join(f"{u} at {l:g}" for l, u in fine.items()))
    for it in range(14):
        W = Th - L - X - S / rho
        Om_prev = Om
        Om, _ = orc.phiplus_stack(W, 1 / rho)
        Th = orc.prox_p(Om + L + X, 0.05 / rho, 0.01 / rho, "FGL")
        C = Th - X - Om
        Lref = orc.rank_stack(C, mu1 / rho)
        rows = []
        for k in range(K):
            B = C[k] - (mu1 / rho) * np.eye(p)
            ev = np.linalg.eigvalsh(B)
            c = 2.4 * np.abs(ev).max()                     # the library's bound sqrt(min(|C^2|_inf, |C^2|_F)) is ~2.4x the radius
            Xc = sign_iterate(B, c, co_c)
            n_unres = int(np.sum(np.abs(ev) < l0c * c))
            L0 = 0.5 * sym(B @ (np.eye(p) + Xc))
            Ld, r, leak, tr = deflate(B, Xc, q, rng)
            rows.append((n_unres, r, np.abs(L0 - Lref[k]).max(), np.abs(Ld - Lref[k]).max(), leak, abs(tr - round(tr)),
                         np.abs(ev).min() / c))
        rows = np.array(rows)
        print(f"it {it:2d} rho {rho:4.2f}: unresolved eigenvalues max {int(rows[:, 0].max())} (detected rank max {int(rows[:, 1].max())}); "
              f"min gap {rows[:, 6].min():.1e}; err(L) coarse only {rows[:, 2].max():.1e} -> deflated {rows[:, 3].max():.1e}; "
              f"probe leak {rows[:, 4].max():.1e}; |trace - integer| {rows[:, 5].max():.1e}", flush=True)
        L = Lref
        X = X + Om - Th + L
        r_, s_, ep, ed = orc.ADMM_stopping_criterion(Om, Om_prev, Th, L, X, S, rho, 1e-20, 1e-20, True)
        rn = 2 * rho if r_ >= 10 * s_ else (0.5 * rho if s_ >= 10 * r_ else rho)
        X = (rho / rn) * X
        rho = rn


if __name__ == "__main__":
    main()
