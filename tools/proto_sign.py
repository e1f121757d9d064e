"""CPU prototype: L-step (C - mu)_+ via scaled Newton-Schulz sign iteration with symmetric products."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ggl_oracle as orc
from gglasso_amd import synth

def up(M): return np.triu(M) + np.triu(M, 1).T

def schedule(l0, maxit=60):
    al = []; l = l0
    for _ in range(maxit):
        a = np.sqrt(3.0 / (1.0 + l + l * l)) if l < 0.99 else 1.0
        al.append(a)
        gl = 0.5 * a * l * (3 - a * a * l * l); g1 = 0.5 * a * (3 - a * a)
        l = min(gl, g1)
        if 1 - l < 4e-16: break
    return al

def rank_ns(C, mu, l0):
    p = C.shape[0]
    B = C - mu * np.eye(p)
    nb = min(np.abs(B).sum(1).max(), np.linalg.norm(B)) * (1 + 1e-10)
    X = B / nb
    al = schedule(l0)
    I = np.eye(p)
    for a in al:
        T = 1.5 * I - 0.5 * a * a * up(X.T @ X)
        X = a * up(X.T @ T)
    resid = np.abs(up(X.T @ X) - I).max()
    L = 0.5 * up(B.T @ X) + 0.5 * B
    return L, len(al), resid

if __name__ == "__main__":
    reg, K, p = "FGL", 4, 120
    S, _ = synth.make_problem(reg, K, p, seed=4)
    Om = np.stack([np.eye(p)] * K); Th = Om.copy(); X = np.zeros_like(S); L = np.zeros_like(S)
    rho = 1.0; mu1 = 0.1
    for it in range(30):
        W = Th - L - X - S / rho
        Om_prev = Om
        Om, _ = orc.phiplus_stack(W, 1 / rho)
        Th = orc.prox_p(Om + L + X, 0.05 / rho, 0.01 / rho, reg)
        C = Th - X - Om
        Lref = orc.rank_stack(C, mu1 / rho)
        D = np.linalg.eigvalsh(C) - mu1 / rho
        nb = np.abs(D).max(axis=1)
        gap = (np.abs(D).min(axis=1) / nb).min()
        rank = (D > 0).sum(axis=1)
        out = []
        for l0 in (1e-2, 1e-4, 1e-6, 1e-9):
            errs = []; res = []; n = 0
            for k in range(K):
                Lk, n, r = rank_ns(C[k], mu1 / rho, l0)
                errs.append(np.abs(Lk - Lref[k]).max()); res.append(r)
            out.append(f"l0={l0:.0e}: n={n} err={max(errs):.1e} resid={max(res):.1e}")
        print(f"it {it:2d} rho {rho:4.2f} rank {rank} relgap {gap:.1e} | " + " | ".join(out))
        L = Lref
        X = X + Om - Th + L
        r, s, ep, ed = orc.ADMM_stopping_criterion(Om, Om_prev, Th, L, X, S, rho, 1e-20, 1e-20, True)
        rn = 2 * rho if r >= 10 * s else (0.5 * rho if s >= 10 * r else rho)
        X = (rho / rn) * X; rho = rn
