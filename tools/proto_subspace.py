#!/usr/bin/env python3
"""VERDICT r4 item 1 -- L-step by warm-started block subspace iteration.  CPU prototype (NumPy) on the iterates of the C4
problem (FGL, p = 500, latent, mu1 = 0.5; K from argv -- the instances are independent in the L-step).

prox_rank_norm (solver/ggl_helper.py:29-36): L = Q diag(max(d - beta, 0)) Q^T of C = Theta - X - Omega, beta = mu1 / rho.
Only the r eigenvalues above beta matter (r <= 6 of 500 on this problem), and C moves little between ADMM iterations.

    V (p x b, b = 16) kept from the previous ADMM iteration;  per pass  V <- orth(f(C) V)  with f = identity on C + sigma I
    (plain subspace iteration) or a Chebyshev filter of degree m that damps [lambda_min, cut] (cut below the b-th Ritz value);
    Rayleigh-Ritz  H = V^T C V = U diag(theta) U^T;  L~ = (V U) diag(max(theta - beta, 0)) (V U)^T.
    Error bound: |L - L~|_F <= sqrt(2) |R_+|_F, R_+ = C V_+ - V_+ Theta_+ the residual of the Ritz pairs above beta, PROVIDED
    the complement holds no eigenvalue above beta (Lipschitz matrix function).

Reported per ADMM iteration (max over the instances): r, the gap structure around beta, passes over C needed for
max|L~ - L| <= 1e-10 cold (random start) and warm (previous V), plain and filtered, and what the residual bound says.

    python tools/proto_subspace.py [K] [p] [iters] [b] [mu1]        (C4's problem)
    python tools/proto_subspace.py --sgl                            (a genuinely low-rank latent problem: fixture G18's data)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth
from oracle import ggl_oracle as orc


def orth(Y):
    Q, _ = np.linalg.qr(Y)
    return Q


def ritz(C, V, beta):
    CV = C @ V
    H = V.T @ CV
    H = 0.5 * (H + H.T)
    th, U = np.linalg.eigh(H)
    Vr = V @ U
    R = CV @ U - Vr * th
    keep = th > beta
    Lt = (Vr[:, keep] * (th[keep] - beta)) @ Vr[:, keep].T
    return Lt, th, Vr, np.linalg.norm(R, axis=0)


def cheb_apply(C, V, lo, cut, m):
    """T_m((2 C - (cut + lo) I) / (cut - lo)) V by the three-term recurrence: m passes over C."""
    e, c = (cut - lo) / 2, (cut + lo) / 2
    Y0 = V
    Y1 = (C @ V - c * V) / e
    for _ in range(m - 1):
        Y0, Y1 = Y1, 2 * (C @ Y1 - c * Y1) / e - Y0
    return Y1


def solve(C, beta, Lref, V0, mode, max_pass=200, tol=1e-10, sigma=None, lo=None):
    """passes over C until max|L~ - Lref| <= tol; returns (passes, V, final err, residual bound)."""
    V = orth(V0)
    passes = 0
    b = V.shape[1]
    while True:
        Lt, th, Vr, rn = ritz(C, V, beta)               # (the Ritz step re-uses C V of the pass before it on the device)
        err = np.abs(Lt - Lref).max()
        bound = np.sqrt(2) * np.linalg.norm(rn[th > beta])
        if err <= tol or passes >= max_pass:
            return passes, Vr, err, bound, th
        if mode == "plain":
            V = orth(C @ Vr + sigma * Vr)
            passes += 1
        else:
            m = mode
            cut = th[0] - 0.02 * (th[-1] - th[0])      # a little below the smallest Ritz value of the block
            cut = max(cut, lo + 1e-3 * abs(lo))
            V = orth(cheb_apply(C, Vr, lo, cut, m))
            passes += m


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    b = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    rng = np.random.default_rng(5)
    S, _ = synth.make_problem("FGL", K, p, N=2 * p, seed=1237)
    Om = np.stack([np.eye(p)] * K); Th = Om.copy(); X = np.zeros_like(S); L = np.zeros_like(S)
    rho, mu1 = 1.0, float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
    Vw = {md: [rng.standard_normal((p, b)) for _ in range(K)] for md in ("plain", 4, 8)}
    print(f"FGL K={K} p={p} latent, mu1={mu1}, block b={b}; passes over C for max|L~ - L| <= 1e-10")
    for it in range(iters):
        W = Th - L - X - S / rho
        Om_prev = Om
        Om, _ = orc.phiplus_stack(W, 1 / rho)
        Th = orc.prox_p(Om + L + X, 0.05 / rho, 0.01 / rho, "FGL")
        C = Th - X - Om
        beta = mu1 / rho
        Lref = orc.rank_stack(C, beta)
        rows = []
        for k in range(K):
            ev = np.linalg.eigvalsh(C[k])[::-1]
            r = int(np.sum(ev > beta))
            sigma = -ev[-1]
            out = {}
            for md in ("plain", 4, 8):
                cold = solve(C[k], beta, Lref[k], rng.standard_normal((p, b)), md, sigma=sigma, lo=ev[-1])
                warm = solve(C[k], beta, Lref[k], Vw[md][k], md, sigma=sigma, lo=ev[-1])
                Vw[md][k] = warm[1]
                out[md] = (cold[0], warm[0], warm[2], warm[3])
            rows.append((r, ev[0], ev[r - 1] - beta if r else np.nan, beta - ev[r], beta - ev[b - 1], beta - ev[b], ev[-1],
                         out["plain"][0], out["plain"][1], out[4][0], out[4][1], out[8][0], out[8][1], out[8][2], out[8][3]))
        a = np.array(rows)
        print(f"it {it:2d} rho {rho:4.2f} beta {beta:.3f}: r {int(a[:,0].min())}..{int(a[:,0].max())}  lam_max {a[:,1].max():.3f} lam_min {a[:,6].min():.3f}  "
              f"gap above {np.nanmin(a[:,2]):.1e} below {a[:,3].min():.1e}  beta-lam_b {a[:,4].min():.2e} beta-lam_b+1 {a[:,5].min():.2e} | "
              f"plain cold {int(a[:,7].max())} warm {int(a[:,8].max())} | cheb4 cold {int(a[:,9].max())} warm {int(a[:,10].max())} | "
              f"cheb8 cold {int(a[:,11].max())} warm {int(a[:,12].max())}  err {a[:,13].max():.1e} bound {a[:,14].max():.1e}", flush=True)
        L = Lref
        X = X + Om - Th + L
        r_, s_, ep, ed = orc.ADMM_stopping_criterion(Om, Om_prev, Th, L, X, S, rho, 1e-20, 1e-20, True)
        rn = 2 * rho if r_ >= 10 * s_ else (0.5 * rho if s_ >= 10 * r_ else rho)
        X = (rho / rn) * X
        rho = rn


def main_sgl():
    """The same measurement on a problem whose latent component IS of low rank: fixture G18's data (the marginal of a Gaussian
    with 5 hidden variables, p = 200, generated by the real reference; ADMM_SGL, lambda1 = 0.03, latent), three mu1."""
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                             "g18_latent_rank_large_p.npz"))
    S = g["sgl_S"]
    p = S.shape[0]
    rng = np.random.default_rng(1)
    b = 16
    for mu1 in (1.5, 0.8, 0.4):
        Om = np.eye(p); Th = Om.copy(); X = np.zeros_like(S); L = np.zeros_like(S); rho = 1.0; lam = 0.03
        Vw = {md: rng.standard_normal((p, b)) for md in ("plain", 4, 8)}
        print(f"ADMM_SGL p={p} latent (G18 data), mu1={mu1}, block b={b}: (cold, warm) passes over C for max|L~ - L| <= 1e-10")
        for it in range(40):
            W = Th - L - X - S / rho
            Omp = Om
            Om = orc.phiplus_stack(W[None], 1 / rho)[0][0]
            Th = orc.prox_od_1norm(Om + L + X, lam / rho)
            C = Th - X - Om
            beta = mu1 / rho
            Lref = orc.rank_stack(C[None], beta)[0]
            ev = np.linalg.eigvalsh(C)[::-1]
            r = int((ev > beta).sum())
            out = {}
            for md in ("plain", 4, 8):
                cold = solve(C, beta, Lref, rng.standard_normal((p, b)), md, sigma=-ev[-1], lo=ev[-1])
                warm = solve(C, beta, Lref, Vw[md], md, sigma=-ev[-1], lo=ev[-1])
                Vw[md] = warm[1]
                out[md] = (cold[0], warm[0], warm[2], warm[3])
            if it < 10 or it % 5 == 0:
                print(f"it {it:2d} rho {rho:4.2f} r {r} lam_max {ev[0]:.3f} lam_min {ev[-1]:.3f} gap above "
                      f"{(ev[r - 1] - beta) if r else np.nan:.1e} below {beta - ev[r]:.1e} beta-lam_b+1 {beta - ev[b]:.1e} | "
                      f"plain {out['plain'][:2]} cheb4 {out[4][:2]} cheb8 {out[8][:2]} err {out[8][2]:.1e} bound {out[8][3]:.1e}",
                      flush=True)
            L = Lref
            X = X + Om - Th + L
            n_r = np.linalg.norm(Om - Th + L)
            n_s = rho * np.linalg.norm(Om - Omp)
            rn = 2 * rho if n_r >= 10 * n_s else (0.5 * rho if n_s >= 10 * n_r else rho)
            X = (rho / rn) * X
            rho = rn


if __name__ == "__main__":
    main_sgl() if "--sgl" in sys.argv else main()
