"""Seedable synthetic covariance stacks for the benchmark and the size-sweep tests (NumPy only; the
GPU box has no /root/reference and needs no networkx).

The construction follows the statistical recipe of the reference's generators
(helper/data_generation.py:14-122 ``generate_precision_matrix(style='erdos')``, :164-195
``group_power_network``, :125-162 ``time_varying_power_network``, :209-236
``sample_covariance_matrix``) -- block-diagonal Erdos-Renyi precision, one block removed per instance
(GGL) or a block swap at half time plus an exponentially decaying block (FGL) -- but is an independent
implementation with its own random stream, so it is statistically equivalent, not bit-equal.  Tests
therefore always feed the SAME S to the oracle and to the HIP path.
"""
import numpy as np


def _blocks(p, M):
    M = min(M, p)
    while p % M:
        M -= 1
    return M, p // M


def base_precision(p, M=10, prob=0.1, rng=None):
    """Sparse SPD precision with M diagonal Erdos-Renyi blocks, unit diagonal."""
    rng = np.random.default_rng() if rng is None else rng
    M, Lb = _blocks(p, M)
    A = np.zeros((p, p))
    for m in range(M):
        adj = np.triu(rng.random((Lb, Lb)) < prob, 1)
        adj = adj | adj.T
        w = rng.uniform(0.1, 0.4, (Lb, Lb)) * rng.choice([-1.0, 1.0], (Lb, Lb))
        A[m * Lb:(m + 1) * Lb, m * Lb:(m + 1) * Lb] = adj * w
    A = A / (1.5 * np.abs(A).sum(axis=1) + 1e-10)[:, None]
    A = 0.5 * (A + A.T) + np.eye(p)
    dmin = np.linalg.eigvalsh(A).min()
    if dmin < 1e-8:
        A += (0.1 + abs(dmin)) * np.eye(p)
    return A, M, Lb


def make_precisions(reg, K, p, M=10, seed=0):
    """(K,p,p) true precision matrices for a Group ('GGL'), Fused ('FGL') or single ('SGL') problem."""
    rng = np.random.default_rng(seed)
    A, M, Lb = base_precision(p, M, rng=rng)
    Theta = np.repeat(A[None], K, axis=0)
    if reg == 'GGL' and K > 1:
        gone = rng.integers(M, size=K)
        for k in range(K):
            b = gone[k]
            blk = slice(b * Lb, (b + 1) * Lb)
            Theta[k, blk, blk] = np.eye(Lb)
    elif reg == 'FGL' and M >= 3:
        decay = np.exp(-0.5 * np.arange(K))
        for k in range(K):
            b = 1 if k <= K / 2 else 0
            blk = slice(b * Lb, (b + 1) * Lb)
            Theta[k, blk, blk] = np.eye(Lb)
            b3 = slice(2 * Lb, 3 * Lb)
            T3 = Theta[k, b3, b3] * decay[k]
            np.fill_diagonal(T3, 1.0)
            Theta[k, b3, b3] = T3
    return Theta


def sample_covariance(Theta, N, seed=0):
    """Biased sample covariance of N Gaussian draws per instance (precision Theta_k)."""
    rng = np.random.default_rng(seed + 7919)
    K, p, _ = Theta.shape
    S = np.empty((K, p, p))
    for k in range(K):
        # x = L^-T z has covariance Theta^-1 when Theta = L L^T
        Lc = np.linalg.cholesky(Theta[k])
        Z = rng.standard_normal((p, N))
        Xs = np.linalg.solve(Lc.T, Z)
        Xs -= Xs.mean(axis=1, keepdims=True)
        S[k] = (Xs @ Xs.T) / N
        S[k] = 0.5 * (S[k] + S[k].T)
    return S


def make_problem(reg, K, p, N=None, seed=0, M=10):
    """Returns (S, Theta_true), both (K,p,p) float64."""
    N = 2 * p if N is None else N
    Theta = make_precisions(reg, K, p, M=M, seed=seed)
    return sample_covariance(Theta, N, seed=seed), Theta
