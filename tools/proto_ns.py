"""CPU prototype: phiplus via scaled Newton-Schulz square root (no eigendecomposition)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ggl_oracle as orc
from gglasso_amd import synth


def alpha_opt(l):
    # optimal scaling of x in [l,1] before the cubic NS map g(x) = x(3-x^2)/2  (Chen & Chow 2014)
    return np.sqrt(3.0 / (1.0 + l + l * l))


def schedule(l0, tol=1e-16, maxit=40):
    """x-interval [l,1] (x = sqrt(m)); returns list of alphas until 1-l < tol."""
    al = []
    l = l0
    for _ in range(maxit):
        a = alpha_opt(l) if l < 0.999 else 1.0
        al.append(a)
        g = lambda x: 0.5 * a * x * (3 - (a * x) ** 2)
        lo = g(l)
        hi = g(1.0)            # after scaling, max of g on [a l, a] is 1 at ax=1; endpoints give the min
        l = min(lo, hi)
        if 1 - l < tol:
            break
    return al


def phiplus_ns(W, beta, bound="inf", verbose=False):
    p = W.shape[0]
    A = W @ W + 4 * beta * np.eye(p)
    if bound == "inf":
        w2 = min(np.abs(W).sum(axis=1).max(), np.linalg.norm(W))   # >= ||W||_2
    elif bound == "exact":
        w2 = np.abs(np.linalg.eigvalsh(W)).max() * 1.0001
    c = w2 * w2 + 4 * beta
    l0 = np.sqrt(4 * beta / c)                     # lower bound of x = sqrt(eig(A)/c)
    al = schedule(l0)
    Y = A / c
    Z = np.eye(p)
    ng = 1
    for it, a in enumerate(al):
        M = (Z @ Y) if it > 0 else Y
        ng += (it > 0)
        T = 1.5 * np.eye(p) - 0.5 * (a * a) * M
        Y = a * (Y @ T); ng += 1
        if it < len(al) - 1:
            Z = a * (T @ Z) if it > 0 else a * T
            ng += (it > 0)
        Y = 0.5 * (Y + Y.T)
        Z = 0.5 * (Z + Z.T)
    sq = np.sqrt(c) * Y
    return 0.5 * (W + sq), len(al), ng, c / (4 * beta)


if __name__ == "__main__":
    reg, K, p = "GGL", 4, 200
    S, _ = synth.make_problem(reg, K, p, seed=3)
    Om0 = np.stack([np.eye(p)] * K)
    # follow an ADMM trajectory and test the Omega-step at every iteration
    Om, Th, X, L = Om0.copy(), Om0.copy(), np.zeros_like(S), np.zeros_like(S)
    rho = 1.0
    for it in range(25):
        W = Th - X - S / rho
        ref, D = orc.phiplus_stack(W, 1 / rho)
        errs, its, ngs, kap = [], [], [], []
        for k in range(K):
            for bound in ("inf", "exact"):
                o, n, ng, kp = phiplus_ns(W[k], 1 / rho, bound)
                if bound == "inf":
                    errs.append(np.abs(o - ref[k]).max() / np.abs(ref[k]).max()); its.append(n); ngs.append(ng); kap.append(kp)
                else:
                    its.append(-n)
        print(f"it {it:2d} rho {rho:5.2f} |W|2 {np.abs(D).max():7.2f} kappa_bound {max(kap):9.1f} NS iters {its[0]} (exact-bound {-its[1]}) gemms {ngs[0]} relerr {max(errs):.2e}")
        Om_prev = Om; Om = ref
        Th = orc.prox_p(Om + X, 0.05 / rho, 0.01 / rho, reg)
        X = X + Om - Th
        r, s, ep, ed = orc.ADMM_stopping_criterion(Om, Om_prev, Th, L, X, S, rho, 1e-20, 1e-20)
        rn = 2 * rho if r >= 10 * s else (0.5 * rho if s >= 10 * r else rho)
        X = (rho / rn) * X; rho = rn
