#!/usr/bin/env python3
"""VERDICT r3 item 3a -- what would the Omega-step's symmetric products cost as Ozaki-split products on the INT8 matrix
cores?  CPU prototype (NumPy; the integer slice products are emulated with exact float64 dgemms on integer-valued matrices).

The Omega-step is 7 symmetric products at the headline: A' = W^2 + 4 beta I, B' = A'^2, then the
(5,9) Newton-Schulz schedule -- Y1 = M T1 (T1 a quadratic in M = A'/c: no product), M2 = T1 Y1, Q = M2^2 + a M2,
T2 = t4 Q (Q + delta I) + e M2 + f I, Omega = (W + sqrt(c) Y1 T2) / 2.  On the FP64 matrix pipe each costs p^3 flop at
78.6 TF/s peak.  The same chip's v_mfma_i32_16x16x64_i8 runs >= 3944 TOPS (MI355X_MICROARCH.md), 50x the FP64 rate, with
EXACT int32 accumulation (k = 512: 512 * 64 * 64 = 2^21 per slice pair, 8 pairs of one weight still < 2^31).

Error-free split (Ozaki et al.): every operand is scaled into [-1, 1] by a power of two (all operands of the scaled
iteration have spectral norm <= 1 up to a known factor, so ONE scale per matrix is enough -- no row scaling, no pass for
row maxima) and cut into signed digits: a first slice of 6 bits, then 7 bits each (|digit| <= 64 fits int8 with room):
    A = sum_t D_t 2^-(6 + 7 t) + r_A,     |r_A| <= 2^-(7 s_A)          (s_A slices)
The product keeps the slice pairs (t, u) with t + u <= d_max; pairs of equal t + u share their power of two and are
accumulated in ONE int32 accumulator.  n_pairs = number of int8 slice products of the product (= MFMA work in units of one
int8 product of the stack).

What the script does: solves the headline problem (GGL, p = 500; K from argv, default 4 -- the error per instance does not
depend on K) with ADMM where the Omega-step is (i) numpy eigh (the reference's), (ii) the (5,9) schedule in float64
products (what the library runs), (iii) the same schedule with every product Ozaki-split at a given slice budget -- and
reports |Theta - Theta_eigh|_F at convergence (tol = rtol = 1e-10) for each budget, plus the slice-pair count per
Omega-step.  Later steps multiply by T = I + E with |E| small: E is sliced instead of T where that saves slices.

    python tools/proto_ozaki.py [K] [p]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth, _lib
from oracle import ggl_oracle as orc


# ---------------------------------------------------------------------------------------------- slicing
def slices(A, scale, s):
    """signed-digit slices of A / scale (|A / scale| <= 1): list of integer-valued float64 matrices D_t and their weights
    2^-(6 + 7 t) (times scale); every |digit| <= 64."""
    r = A / scale
    out, w = [], []
    for t in range(s):
        wt = 2.0 ** -(6 + 7 * t)
        D = np.rint(r / wt)
        assert np.abs(D).max() <= 65, np.abs(D).max()
        out.append(D)
        w.append(wt)
        r = r - D * wt
    return out, w


PAIRS = 0          # int8 slice products executed (one unit = one full product of the stack on the int8 pipe)


def oz_mul(A, B, sa, sb, scale_a, scale_b, dmax=None):
    """A @ B from sa / sb slices of the operands, slice pairs with t + u <= dmax (default: the triangular truncation
    matching the coarser operand)."""
    global PAIRS
    DA, wa = slices(A, scale_a, sa)
    DB, wb = slices(B, scale_b, sb)
    if dmax is None:
        dmax = max(sa, sb) - 1
    C = np.zeros_like(A)
    for d in range(min(dmax, sa + sb - 2), -1, -1):          # smallest weights first
        acc = None                                          # one int32 accumulator per weight 2^-(12 + 7 d)
        for t in range(sa):
            u = d - t
            if 0 <= u < sb:
                prod = DA[t] @ DB[u]                          # exact: integer-valued, |sum| <= p * 65^2 < 2^53
                acc = prod if acc is None else acc + prod
                PAIRS += 1
        if acc is not None:
            assert np.abs(acc).max() < 2 ** 31
            C += acc * (2.0 ** -(12 + 7 * d))
    return C * (scale_a * scale_b)


def pow2_above(x):
    return 2.0 ** np.ceil(np.log2(x))


# ---------------------------------------------------------------------------------------------- the schedule
def plan(l, tol):
    lib = _lib.load()
    deg = (ctypes.c_int * 24)()
    co = (ctypes.c_double * (24 * 6))()
    units = ctypes.c_int()
    n = lib.ggl_dev_ns_schedule_tol(float(l), 9, float(tol), 24, deg, co, ctypes.byref(units))
    assert n > 0
    return list(deg)[:n], np.array(co[:6 * n]).reshape(n, 6), units.value


def omega_ns(W, beta, mul, budget=None, ns_tol=2e-12):
    """phiplus(W) = (W + (W^2 + 4 beta I)^(1/2)) / 2 by the library's schedule; mul(A, B, tag) is the product routine."""
    p = W.shape[0]
    I = np.eye(p)
    A1 = mul(W, W, "A'") + 4 * beta * I
    B1 = mul(A1, A1, "B'")
    # bound: the library takes sqrt(min(|B'|_inf, Collatz-Wielandt, |B'|_F)); here the exact lambda_max * 1.02 (what the carried
    # vector converges to)
    c = 1.02 * np.sqrt(np.linalg.eigvalsh(B1)[-1])
    l = np.sqrt(4 * beta / c)
    deg, co, units = plan(l, ns_tol)
    M = A1 / c
    Y, Z = M, None            # Z0 = I
    M2pow = B1 / c ** 2       # M^2 of the first step is B'/c^2: no product
    for i, d in enumerate(deg):
        t = co[i]
        last = i == len(deg) - 1
        if i > 0:
            M = mul(Z, Y, f"M{i}")
            M2pow = None
        if d == 3:
            T = t[0] * I + t[1] * M
        elif d == 5:
            if M2pow is None:
                M2pow = mul(M, M, f"M{i}^2")
            T = t[0] * I + t[1] * M + t[2] * M2pow
        else:
            if M2pow is None:
                M2pow = mul(M, M, f"M{i}^2")
            # quartic in M: t0 + t1 m + t2 m^2 + t3 m^3 + t4 m^4 = t0 + t1 m + (t2 + t3 m + t4 m^2) m^2
            U = t[2] * I + t[3] * M + t[4] * M2pow
            T = t[0] * I + t[1] * M + mul(U, M2pow, f"U{i}M^2")
        Ynew = mul(Y, T, f"Y{i}T")
        if not last:
            Z = T if Z is None else mul(T, Z, f"TZ{i}")
        Y = Ynew
    return 0.5 * (W + np.sqrt(c) * Y), units


def exact_mul(A, B, tag):
    C = A @ B
    return 0.5 * (C + C.T)


def omega_oz_smart(W, beta, s_full, cfg, ns_tol=2e-12):
    """The same schedule with the slice budget spent where it is needed.  Full-precision (s_full slices, triangular) products:
    A', B', the first step's products and every M_i = Z Y (where F_i = I - M_i is formed by cancellation).  From the second
    step on everything is a polynomial in the SMALL matrix F = I - M (|F| <= 1 - l_i^2): t(M) = g(F) = g0 + g1 F + F^2 (g2 + g3
    F + g4 F^2), and the update is Y <- g0 Y + Y E, E = g(F) - g0 I -- products of small matrices need few slices relative to
    their own scale.  cfg = (s_F2, s_GF2, s_YE, d_YE): slices for F F, for (g3 F + g4 F^2) F^2, and slices / diagonal cut for Y E."""
    p = W.shape[0]
    I = np.eye(p)
    full = OzMul(s_full)
    A1 = full(W, W, "A'") + 4 * beta * I
    B1 = full(A1, A1, "B'")
    c = 1.02 * np.sqrt(np.linalg.eigvalsh(B1)[-1])
    l = np.sqrt(4 * beta / c)
    deg, co, units = plan(l, ns_tol)
    M = A1 / c
    M2 = B1 / c ** 2
    t = co[0]
    if deg[0] == 3:
        T = t[0] * I + t[1] * M
    elif deg[0] == 5:
        T = t[0] * I + t[1] * M + t[2] * M2
    else:
        T = t[0] * I + t[1] * M + full(t[2] * I + t[3] * M + t[4] * M2, M2, "UM2")
    Y = full(M, T, "Y0T")
    Z = T
    s_F2, s_GF2, s_YE, d_YE = cfg
    for i in range(1, len(deg)):
        last = i == len(deg) - 1
        Mi = full(Z, Y, f"M{i}")
        F = I - Mi
        fs = pow2_above(np.abs(F).max())
        t = co[i]
        # g(f) = t(1 - f): coefficients of the polynomial in f
        tt = np.polynomial.polynomial.Polynomial(t[:5])
        g = tt(np.polynomial.polynomial.Polynomial([1.0, -1.0])).coef
        g = np.concatenate([g, np.zeros(5 - len(g))])
        E = g[1] * F
        if deg[i] >= 5:
            C = oz_mul(F, F, s_F2, s_F2, fs, fs)
            F2 = 0.5 * (C + C.T)
            E = E + g[2] * F2
            if deg[i] == 9:
                G = g[3] * F + g[4] * F2
                C = oz_mul(G, F2, s_GF2, s_GF2, pow2_above(np.abs(G).max()), pow2_above(np.abs(F2).max()))
                E = E + 0.5 * (C + C.T)
        C = oz_mul(Y, E, s_YE, s_YE, pow2_above(np.abs(Y).max()), pow2_above(np.abs(E).max()), dmax=d_YE)
        Ynew = g[0] * Y + 0.5 * (C + C.T)
        if not last:
            C = oz_mul(E, Z, s_YE, s_YE, pow2_above(np.abs(E).max()), pow2_above(np.abs(Z).max()), dmax=d_YE)
            Z = g[0] * Z + 0.5 * (C + C.T)
        Y = Ynew
    return 0.5 * (W + np.sqrt(c) * Y), units


class OzMul:
    """every product Ozaki-split: s slices per operand, triangular truncation; operands scaled by a power of two above
    their largest entry (the kernel would use the a-priori spectral bounds: |entry| <= |.|_2)."""
    def __init__(self, s, s_first=None):
        self.s, self.s_first = s, s_first or s

    def __call__(self, A, B, tag):
        s = self.s_first if tag in ("A'", "B'") else self.s
        C = oz_mul(A, B, s, s, pow2_above(np.abs(A).max()), pow2_above(np.abs(B).max()))
        return 0.5 * (C + C.T)


# ---------------------------------------------------------------------------------------------- ADMM around it
def admm(S, l1, l2, omega_step, tol=1e-10, max_iter=200):
    K, p, _ = S.shape
    Om = np.repeat(np.eye(p)[None], K, axis=0)
    Th, X = Om.copy(), np.zeros_like(Om)
    rho = 1.0
    dim = K * (p * p + p) / 2
    for it in range(max_iter):
        Om_prev = Om
        W = Th - X - S / rho
        Om = np.stack([omega_step(W[k], 1.0 / rho) for k in range(K)])
        Th = orc.prox_p(Om + X, l1 / rho, l2 / rho, "GGL")
        X = X + Om - Th
        r = np.linalg.norm(Om - Th)
        s = rho * np.linalg.norm(Om - Om_prev)
        e_pri = dim * tol + tol * max(np.linalg.norm(Om), np.linalg.norm(Th))
        e_dual = dim * tol + tol * rho * np.linalg.norm(X)
        rho_new = 2 * rho if r >= 10 * s else (0.5 * rho if s >= 10 * r else rho)
        X *= rho / rho_new
        rho = rho_new
        if r <= e_pri and s <= e_dual:
            return Th, it + 1
    return Th, max_iter


def main():
    global PAIRS
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=1239)

    def eigh_step(W, beta):
        d, Q = np.linalg.eigh(W)
        return (Q * (0.5 * (d + np.sqrt(d * d + 4 * beta)))) @ Q.T

    Th_ref, it_ref = admm(S, 0.05, 0.01, eigh_step)
    print(f"GGL K={K} p={p}: eigh Omega-step: {it_ref} iterations, |Theta|_F = {np.linalg.norm(Th_ref):.4g}")
    units = []
    Th, it = admm(S, 0.05, 0.01, lambda W, b: (lambda r: (units.append(r[1]), r[0])[1])(omega_ns(W, b, exact_mul)))
    print(f"float64 products, (5,9) schedule at ns_tol 2e-12: {it} iterations, products per Omega-step "
          f"{np.mean(units):.2f}, |dTheta|_F = {np.linalg.norm(Th - Th_ref):.3e}")
    print(f"{'slices':>7} {'first2':>7} {'int8 slice products / Omega-step':>33} {'per fp64 product':>17} "
          f"{'iterations':>10} {'|dTheta|_F':>11} {'max|dTheta|':>12}")
    for s, s_first in ((6, 6), (7, 7), (8, 8)):
        PAIRS = 0
        calls = []
        mulr = OzMul(s, s_first)

        def step(W, b):
            Om, u = omega_ns(W, b, mulr)
            calls.append(u)
            return Om
        Th, it = admm(S, 0.05, 0.01, step)
        per_step = PAIRS / len(calls)
        print(f"{s:>7} {s_first:>7} {per_step:>33.1f} {per_step / np.mean(calls):>17.1f} {it:>10} "
              f"{np.linalg.norm(Th - Th_ref):>11.3e} {np.abs(Th - Th_ref).max():>12.3e}", flush=True)
    smart(S, Th_ref)


def smart(S, Th_ref):
    global PAIRS
    print("budget where it is needed (polynomials in F = I - M from the second step on):")
    print(f"{'s_full':>7} {'F^2':>4} {'G F^2':>6} {'Y E (s,d)':>10} {'int8 slice products / Omega-step':>33} {'per fp64 product':>17} "
          f"{'iterations':>10} {'|dTheta|_F':>11} {'max|dTheta|':>12}")
    for s_full, cfg in ((7, (4, 3, 5, 4)), (7, (4, 2, 5, 4)), (7, (3, 3, 5, 4)), (7, (4, 3, 4, 4)), (7, (4, 3, 5, 3)),
                        (6, (4, 3, 5, 4)), (7, (3, 2, 4, 3))):
        PAIRS = 0
        calls = []

        def step(W, b):
            Om, u = omega_oz_smart(W, b, s_full, cfg)
            calls.append(u)
            return Om
        Th, it = admm(S, 0.05, 0.01, step)
        per_step = PAIRS / len(calls)
        print(f"{s_full:>7} {cfg[0]:>4} {cfg[1]:>6} {str(cfg[2:]):>10} {per_step:>33.1f} {per_step / np.mean(calls):>17.1f} {it:>10} "
              f"{np.linalg.norm(Th - Th_ref):>11.3e} {np.abs(Th - Th_ref).max():>12.3e}", flush=True)


if __name__ == "__main__":
    main()
