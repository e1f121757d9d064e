"""CPU prototype (dev tool): generic odd-degree minimax step in the shifted basis x (1-x^2)^j, the
factorised evaluation of a degree-9 step with two products, and its accuracy on matrices."""
import numpy as np
from math import comb


def remez_step(l, m, grid=1024):
    """minimax p(x) = x t(x^2) ~ 1 on [l,1], t of degree m-1.  Returns monomial coefficients of t (rescaled so
    max p = 1) and the new lower end."""
    smax = 1 - l * l
    ph = lambda x: np.stack([x * ((1 - x * x) / smax) ** j for j in range(m)], -1)
    k = np.arange(m + 1)
    pts = l + (1 - l) * (1 - np.cos(np.pi * k / m)) / 2
    g = l + (1 - l) * (1 - np.cos(np.pi * np.arange(grid + 1) / grid)) / 2
    G = ph(g)
    for it in range(40):
        A = np.concatenate([ph(pts), -((-1.0) ** (k + 1))[:, None]], 1)
        sol = np.linalg.solve(A, np.ones(m + 1))
        co, E = sol[:m], sol[m]
        err = G @ co - 1
        ext = [0] + [i for i in range(1, grid) if (err[i] - err[i - 1]) * (err[i + 1] - err[i]) <= 0] + [grid]
        alt = []
        for i in ext:
            if alt and np.sign(err[i]) == np.sign(err[alt[-1]]):
                if abs(err[i]) > abs(err[alt[-1]]):
                    alt[-1] = i
            else:
                alt.append(i)
        while len(alt) > m + 1:
            alt.pop(0) if abs(err[alt[0]]) < abs(err[alt[-1]]) else alt.pop()
        if len(alt) < m + 1:
            break
        new = g[alt]
        if np.abs(new - pts).max() < 1e-14:
            break
        pts = new
    # monomial coefficients of t(mm) = sum_j co_j ((1-mm)/smax)^j
    t = np.zeros(m)
    for j in range(m):
        for i in range(j + 1):
            t[i] += co[j] / smax ** j * comb(j, i) * (-1) ** i
    x = g
    p = x * np.polyval(t[::-1], x * x)
    E = np.abs(p - 1).max() * (1 + 1e-3)       # grid maximum, inflated for what lies between grid points
    return t / (1 + E), (1 - E) / (1 + E)


def eval_T9(M, t, S):
    """T = t0 + t1 M + t2 M^2 + t3 M^3 + t4 M^4 with two products: Q = M^2 + a M, T = t4 Q (Q + d I) + e M + f I."""
    I = np.eye(len(M))
    a = t[3] / (2 * t[4]); d = t[2] / t[4] - a * a; e = t[1] - t[4] * d * a; f = t[0]
    Q = S(M @ M) + a * M
    return t[4] * S(Q @ (Q + d * I)) + e * M + f * I, (a, d, e, f)


if __name__ == "__main__":
    for l in [0.5, 0.6, 0.65, 0.67, 0.7, 0.75]:
        t1, l1 = remez_step(l, 5)
        t2, l2 = remez_step(l1, 5)
        print(f"l={l}: after 9: 1-l={1-l1:.3e}; after 9,9: {1-l2:.3e}; coef1 {np.round(t1,3)} coef2 {np.round(t2,4)}")
    rng = np.random.default_rng(0)
    p = 200
    S = lambda M: 0.5 * (M + M.T)
    for kappa in [2, 4, 10, 30]:
        Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
        beta = 1.0
        wmax = np.sqrt(4 * beta * (kappa - 1))
        w = rng.uniform(-wmax, wmax, p); w[0] = wmax
        W = S((Q * w) @ Q.T)
        ref = (Q * (0.5 * (w + np.sqrt(w * w + 4 * beta)))) @ Q.T
        A = W @ W + 4 * beta * np.eye(p); B = A @ A
        c = np.linalg.eigvalsh(A)[-1] * 1.15
        l = np.sqrt(4 * beta / c)
        I = np.eye(p)
        # first step degree 9 from A', B'
        seq = []
        ll = l
        while 1 - ll > 4e-16:
            t, ll = remez_step(ll, 5) if 1 - ll > 1e-7 else (np.array([315, -420, 378, -180, 35]) / 128, 1.0)
            seq.append(t)
        m = A / c
        for i, t in enumerate(seq):
            if i == 0:
                a = t[3] / (2 * t[4]); d = t[2] / t[4] - a * a; e = t[1] - t[4] * d * a; f = t[0]
                Q1 = B / c ** 2 + a * m
                T = t[4] * S(Q1 @ (Q1 + d * I)) + e * m + f * I
                Z = T; Y = S(m @ T)
            else:
                M = S(Z @ Y)
                T, par = eval_T9(M, t, S)
                Y, Z = S(Y @ T), S(T @ Z)
        o = 0.5 * (W + np.sqrt(c) * Y)
        print(f"kappa {kappa} l {l:.3f} steps {len(seq)} relerr {np.abs(o-ref).max()/np.abs(ref).max():.2e}  fact (a,d,e,f) last {np.round(par,3) if len(seq)>1 else None}")
