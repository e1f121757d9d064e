"""CPU prototype: degree sequence {3,5} planner for the coupled Newton-Schulz square root (dev tool).

x in [l,1] is sqrt(eig(Z Y)).  A cubic step maps x -> a x (3 - a^2 x^2)/2 (Chen-Chow scaling), a quintic
step x -> x (q0 + q1 x^2 + q2 x^4) with the minimax coefficients on [l,1] (Remez; the interior extrema are
roots of a quadratic in x^2).  Cost in symmetric products: cubic middle 3 / last 2 / first 0,
quintic middle 4 / last 3 / first 1 (on top of the two products A', B')."""
import itertools
import numpy as np


def cubic_coef(l):
    a = np.sqrt(3.0 / (1.0 + l + l * l)) if l < 0.99 else 1.0
    # t(m) = 1.5 a - 0.5 a^3 m
    gl = 0.5 * a * l * (3 - a * a * l * l)
    g1 = 0.5 * a * (3 - a * a)
    return (1.5 * a, -0.5 * a ** 3, 0.0), min(gl, g1)


def quintic_coef(l):
    if 1 - l < 1e-3:
        e = 1 - l
        return (15 / 8, -10 / 8, 3 / 8), 1 - 2.5 * e ** 3 - 1e-17
    q1, q2 = l + (1 - l) / 3, l + 2 * (1 - l) / 3      # interior extrema guess
    for _ in range(100):
        pts = np.array([l, q1, q2, 1.0])
        sg = np.array([-1.0, 1.0, -1.0, 1.0])
        A = np.stack([pts, pts ** 3, pts ** 5, -sg], 1)
        a, b, c, E = np.linalg.solve(A, np.ones(4))
        # p' = a + 3 b x^2 + 5 c x^4 = 0
        disc = 9 * b * b - 20 * a * c
        r = np.sort([(-3 * b - np.sqrt(disc)) / (10 * c), (-3 * b + np.sqrt(disc)) / (10 * c)])
        n1, n2 = np.sqrt(r[0]), np.sqrt(r[1])
        if abs(n1 - q1) + abs(n2 - q2) < 1e-15:
            break
        q1, q2 = n1, n2
    s = 1.0 / (1.0 + E)          # rescale so the maximum is 1
    return (a * s, b * s, c * s), (1 - E) / (1 + E)


def plan(l0, tol=4e-16, maxn=12):
    best = None
    for n in range(1, maxn):
        for seq in itertools.product((3, 5), repeat=n):
            l = l0
            for d in seq:
                _, l = (cubic_coef if d == 3 else quintic_coef)(l)
            if 1 - l > tol:
                continue
            cost = 2
            for i, d in enumerate(seq):
                first, last = i == 0, i == n - 1
                if d == 3:
                    cost += 0 if first else (2 if last else 3)
                else:
                    cost += 1 if first else (3 if last else 4)
                if first and last and d == 5:
                    pass
            if best is None or cost < best[0]:
                best = (cost, seq)
        if best is not None and n > len(best[1]) + 1:
            break
    return best


def run(W, beta, seq, c=None, sym=True):
    p = W.shape[0]
    I = np.eye(p)
    A = W @ W + 4 * beta * I
    B = A @ A
    if c is None:
        c = np.sqrt(min(np.abs(B).sum(1).max(), np.linalg.norm(B)))
    l = np.sqrt(4 * beta / c)
    S = (lambda M: 0.5 * (M + M.T)) if sym else (lambda M: M)
    Y = Z = None
    for i, d in enumerate(seq):
        co, l = (cubic_coef if d == 3 else quintic_coef)(l)
        if i == 0:
            T = co[0] * I + co[1] * (A / c) + co[2] * (B / c ** 2)
            if d == 3:
                Y = co[0] * A / c + co[1] * B / c ** 2
            else:
                Y = S((A / c) @ T)
            Z = T
            continue
        M = S(Z @ Y)
        if d == 3:
            T = co[0] * I + co[1] * M
        else:
            T = co[0] * I + co[1] * M + co[2] * S(M @ M)
        Y, Z = S(Y @ T), S(T @ Z)
    return 0.5 * (W + np.sqrt(c) * Y)


if __name__ == "__main__":
    for l0 in [0.9, 0.5, 0.3, 0.2, 0.15, 0.1, 0.07, 0.058, 0.03, 0.01, 1e-3]:
        cb = plan(l0)
        # cubic only
        l, n = l0, 0
        while 1 - l > 4e-16:
            _, l = cubic_coef(l); n += 1
        print(f"l0={l0:g} kappa={1/l0**2:.4g}: cubic n={n} cost={3*n-2 if n>1 else 2}   best {cb}")
    rng = np.random.default_rng(0)
    p = 200
    for kappa in [2, 10, 30, 100, 300, 1000]:
        Q, _ = np.linalg.qr(rng.standard_normal((p, p)))
        beta = 1.0
        wmax = np.sqrt(4 * beta * (kappa - 1))
        w = rng.uniform(-wmax, wmax, p); w[0] = wmax; w[1] = -wmax
        W = (Q * w) @ Q.T; W = 0.5 * (W + W.T)
        ref = (Q * (0.5 * (w + np.sqrt(w * w + 4 * beta)))) @ Q.T
        A = W @ W + 4 * beta * np.eye(p); B = A @ A
        c = np.sqrt(min(np.abs(B).sum(1).max(), np.linalg.norm(B)))
        l0 = np.sqrt(4 * beta / c)
        cost, seq = plan(l0)
        l, n = l0, 0
        while 1 - l > 4e-16:
            _, l = cubic_coef(l); n += 1
        for s in (seq, (3,) * n):
            o = run(W, beta, s, c)
            print(f"kappa {kappa} l0 {l0:.3f} seq {s}: relerr {np.abs(o-ref).max()/np.abs(ref).max():.2e}")
