"""Operator seam: the reference's prox / eigen operators with identical signatures, computed by
the gfx950 kernels through the C ABI (host NumPy arrays in, NumPy arrays out).

Reference (paths relative to /root/reference/src/gglasso/): solver/ggl_helper.py, solver/fgl_helper.py.
These entry points exist for operator-level parity and for callers that use single operators; the
ADMM loop itself keeps its state on the device (see solver.py)."""
import numpy as np

from . import _lib
from ._lib import as_c, check, ptr

_REG = {"GGL": _lib.REG_GGL, "FGL": _lib.REG_FGL}


def _lib_gpu():
    _lib.require_gpu()
    return _lib.load()


def eigh(A, method=_lib.EIG_AUTO):
    """numpy.linalg.eigh for a (p,p) matrix or (K,p,p) stack (lower triangle read, ascending
    eigenvalues, eigenvectors in columns) -- call sites solver/admm_solver.py:181,199."""
    A = as_c(A)
    single = A.ndim == 2
    A3 = A[None] if single else A
    K, p, p2 = A3.shape
    assert p == p2
    D = np.empty((K, p))
    Q = np.empty((K, p, p))
    check(_lib_gpu().ggl_eigh_batched(K, p, ptr(A3), ptr(D), ptr(Q), method))
    return (D[0], Q[0]) if single else (D, Q)


def phiplus(beta, D, Q):
    """solver/ggl_helper.py:280-303: prox of -beta*logdet from an eigendecomposition.
    (p,) / (p,p) inputs as in the reference, or stacks (K,p) / (K,p,p) with beta scalar or (K,)."""
    D, Q = as_c(D), as_c(Q)
    single = Q.ndim == 2
    D3, Q3 = (D[None], Q[None]) if single else (D, Q)
    K, p, _ = Q3.shape
    b = as_c(np.broadcast_to(np.asarray(beta, dtype=np.float64), (K,)))
    out = np.empty((K, p, p))
    check(_lib_gpu().ggl_phiplus(K, p, ptr(b), ptr(as_c(D3)), ptr(as_c(Q3)), ptr(out)))
    return out[0] if single else out


def prox_rank_norm(A, beta, D=np.array([]), Q=np.array([])):
    """solver/ggl_helper.py:29-36.  As in the reference, the eigendecomposition is computed here
    when D does not match A (the reference prints a notice in that case; we stay silent)."""
    A = as_c(A)
    single = A.ndim == 2
    A3 = A[None] if single else A
    K, p, _ = A3.shape
    b = as_c(np.broadcast_to(np.asarray(beta, dtype=np.float64), (K,)))
    out = np.empty((K, p, p))
    D = np.asarray(D)
    if D.shape[-1:] != (p,) or D.size != K * p:
        check(_lib_gpu().ggl_rank_matrix(K, p, ptr(b), ptr(A3), ptr(out), _lib.EIG_AUTO))
    else:
        D3 = as_c(D.reshape(K, p))
        Q3 = as_c(np.asarray(Q).reshape(K, p, p))
        check(_lib_gpu().ggl_prox_rank_norm(K, p, ptr(b), ptr(D3), ptr(Q3), ptr(out)))
    return out[0] if single else out


def phiplus_matrix(W, beta, method=_lib.EIG_AUTO, ns_mode=0, ns_degrees=0):
    """eigh + phiplus fused (what one Omega-step runs, admm_solver.py:180-187).  ns_mode / ns_degrees: the
    Newton-Schulz controls of include/ggl_hip.h (GGL_EIG_NS_MODE / GGL_EIG_NS_DEGREES), for the parity tests."""
    W = as_c(W)
    single = W.ndim == 2
    W3 = W[None] if single else W
    K, p, _ = W3.shape
    b = as_c(np.broadcast_to(np.asarray(beta, dtype=np.float64), (K,)))
    out = np.empty((K, p, p))
    check(_lib_gpu().ggl_phiplus_matrix(K, p, ptr(b), ptr(W3), ptr(out), _lib.eig_flags(method, ns_mode, ns_degrees)))
    return out[0] if single else out


def rank_matrix(C, beta, method=_lib.EIG_AUTO, ns_degrees=0):
    """eigh + prox_rank_norm fused (L-step, admm_solver.py:197-205)."""
    C = as_c(C)
    single = C.ndim == 2
    C3 = C[None] if single else C
    K, p, _ = C3.shape
    b = as_c(np.broadcast_to(np.asarray(beta, dtype=np.float64), (K,)))
    out = np.empty((K, p, p))
    check(_lib_gpu().ggl_rank_matrix(K, p, ptr(b), ptr(C3), ptr(out), _lib.eig_flags(method, 0, ns_degrees)))
    return out[0] if single else out


def prox_od_1norm(A, l):
    """solver/ggl_helper.py:16-27; l is a scalar or a (p,p) array."""
    A = as_c(A)
    assert A.ndim == 2 and A.shape[0] == A.shape[1]
    p = A.shape[0]
    out = np.empty((p, p))
    if np.ndim(l) == 0:
        check(_lib_gpu().ggl_prox_od_1norm(p, ptr(A), float(l), None, ptr(out)))
    else:
        lam = as_c(np.broadcast_to(l, (p, p)))
        check(_lib_gpu().ggl_prox_od_1norm(p, ptr(A), 0.0, ptr(lam), ptr(out)))
    return out


def prox_p(X, l1, l2, reg):
    """solver/ggl_helper.py:190-207 (same asserts: symmetric input to 1e-5, positive lambdas)."""
    X = as_c(X)
    assert np.abs(X - X.transpose(0, 2, 1)).max() <= 1e-5, "input X is not symmetric"
    assert min(l1, l2) > 0, "lambda 1 and lambda2 have to be positive"
    assert reg in ('GGL', 'FGL')
    K, p, _ = X.shape
    out = np.empty((K, p, p))
    check(_lib_gpu().ggl_prox_p(K, p, ptr(X), float(l1), float(l2), _REG[reg], ptr(out)))
    return out


def _vec(fn, v, *scalars):
    v = as_c(v)
    single = v.ndim == 1
    Y = v[None] if single else v
    n, K = Y.shape
    out = np.empty((n, K))
    check(fn(n, K, ptr(Y), *scalars, ptr(out)))
    return out[0] if single else out


def prox_tv(v, l):
    """solver/ggl_helper.py:126-129 == condat_method (solver/fgl_helper.py:11-68).  v is (K,) or a
    batch (n,K) of independent signals."""
    return _vec(_lib_gpu().ggl_prox_tv, v, float(l))


condat_method = prox_tv


def prox_2norm(v, l):
    """solver/ggl_helper.py:38-43 for a vector (K,) or a batch (n,K)."""
    return _vec(_lib_gpu().ggl_prox_2norm, v, float(l))


def prox_phi_ggl(v, l1, l2):
    """solver/ggl_helper.py:68-71."""
    return _vec(_lib_gpu().ggl_prox_phi, v, float(l1), float(l2), _lib.REG_GGL)


def prox_phi_fgl(v, l1, l2):
    """solver/ggl_helper.py:131-134."""
    return _vec(_lib_gpu().ggl_prox_phi, v, float(l1), float(l2), _lib.REG_FGL)
