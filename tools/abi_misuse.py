#!/usr/bin/env python3
"""C-ABI misuse probe (include/ggl_hip.h): arguments a careless binding could pass -- K or p of 0 or negative, a device that does not
exist, NULL handles and buffers, instance indices out of range, unknown options / penalties, non-positive rho / lambda, steps on a
ctx that was never given S -- must come back as an error CODE with a message in ggl_last_error, never as a crash, a hang or a
silent success.  Every call runs in the same process; the script ends with a normal solve on a fresh ctx to show the library is
still in order.        python tools/abi_misuse.py        -> one line per probe, 'ok' at the end"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib  # noqa: E402

lib = _lib.load()
NULL = None
bad = 0


def last():
    try:
        lib.ggl_last_error.restype = ctypes.c_char_p
        return (lib.ggl_last_error() or b"").decode()[:90]
    except Exception:
        return ""


def expect_error(name, rc):
    global bad
    ok = rc != 0
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} {name}: rc {rc} {last() if ok else '(accepted!)'}", flush=True)


def ctx(K, p, device=0, flags=0):
    h = ctypes.c_void_p()
    rc = lib.ggl_ctx_create(device, K, p, flags, None, ctypes.byref(h))
    return rc, h


for K, p in ((0, 8), (-1, 8), (4, 0), (4, -3), (0, 0)):
    rc, h = ctx(K, p)
    expect_error(f"ggl_ctx_create(K={K}, p={p})", rc)
rc, h = ctx(2, 8, device=99)
expect_error("ggl_ctx_create(device=99)", rc)
expect_error("ggl_ctx_create(out=NULL)", lib.ggl_ctx_create(0, 2, 8, 0, None, None))
expect_error("ggl_ctx_sync(NULL)", lib.ggl_ctx_sync(None))
rc = lib.ggl_ctx_destroy(None)            # like free(NULL): a no-op that succeeds
print(f"{'ok ' if rc == 0 else 'BAD'} ggl_ctx_destroy(NULL): rc {rc} (a no-op)")
bad += 0 if rc == 0 else 1
expect_error("ggl_ctx_set_option(NULL, ...)", lib.ggl_ctx_set_option(None, 1, ctypes.c_double(0.0)))

rc, h = ctx(3, 12)
assert rc == 0, last()
dbl = ctypes.POINTER(ctypes.c_double)
S = np.stack([np.eye(12)] * 3)
ptr = lambda a: a.ctypes.data_as(dbl)
expect_error("ggl_ctx_set_option(option=9999)", lib.ggl_ctx_set_option(h, 9999, ctypes.c_double(1.0)))
expect_error("ggl_ctx_get_option(option=-1)", lib.ggl_ctx_get_option(h, -1, ctypes.byref(ctypes.c_double())))
expect_error("ggl_ctx_get_option(value=NULL)", lib.ggl_ctx_get_option(h, 1, None))
expect_error("ggl_set_S(NULL buffer)", lib.ggl_set_S(h, None))
# (the failed hipSetDevice(99) above must not be what the next valid call reports: the runtime keeps a failed call's code as the
# thread's last error until somebody takes it)
rc = lib.ggl_set_S_ex(h, ptr(S), 0)
print(f"{'ok ' if rc == 0 else 'BAD'} ggl_set_S_ex(period=0: all K) after the failed ggl_ctx_create: rc {rc} {last() if rc else ''}")
bad += 0 if rc == 0 else 1
expect_error("ggl_set_S_ex(period=5 of K=3)", lib.ggl_set_S_ex(h, ptr(S), 5))
expect_error("ggl_set_S_ex(period=2 of K=3)", lib.ggl_set_S_ex(h, ptr(S), 2))
expect_error("ggl_set_S_ex(period=-1)", lib.ggl_set_S_ex(h, ptr(S), -1))
expect_error("ggl_get_state_k(k=3 of K=3)", lib.ggl_get_state_k(h, 3, ptr(S), None, None, None))
expect_error("ggl_get_state_k(k=-1)", lib.ggl_get_state_k(h, -1, ptr(S), None, None, None))
norms = np.zeros(8)
assert lib.ggl_set_S(h, ptr(S)) == 0 and lib.ggl_set_state(h, ptr(S), ptr(S), None, ptr(np.zeros_like(S))) == 0, last()
step = lambda rho, l1, l2, reg, latent=0: lib.ggl_admm_step(h, ctypes.c_double(rho), ctypes.c_double(l1), ctypes.c_double(l2), reg, latent,
                                                            None, None, ptr(norms))
expect_error("ggl_admm_step(rho=0)", step(0.0, 0.1, 0.1, 0))
expect_error("ggl_admm_step(rho=-1)", step(-1.0, 0.1, 0.1, 0))
expect_error("ggl_admm_step(rho=nan)", step(float("nan"), 0.1, 0.1, 0))
expect_error("ggl_admm_step(reg=7)", step(1.0, 0.1, 0.1, 7))
expect_error("ggl_admm_step(latent without mu1)", step(1.0, 0.1, 0.1, 0, latent=1))
expect_error("ggl_snapshot_k(k=17)", lib.ggl_snapshot_k(h, 17))
expect_error("ggl_threshold_scan(ntau=0)", lib.ggl_threshold_scan(h, ptr(norms), 0, ptr(norms), None))
expect_error("ggl_ext_setup(NULL)", lib.ggl_ext_setup(h, None, None, 0))
expect_error("ggl_set_instance_dims(pk > p)", lib.ggl_set_instance_dims(h, (ctypes.c_int * 3)(4, 13, 5)))
expect_error("ggl_set_instance_dims(pk = 0)", lib.ggl_set_instance_dims(h, (ctypes.c_int * 3)(4, 0, 5)))
rho3, lam3 = np.ones(3), np.full(3, 0.1)
expect_error("ggl_sgl_batch_step(rho=NULL)", lib.ggl_sgl_batch_step(h, None, ptr(lam3), 0, None, ptr(np.zeros(15))))
rho_bad = np.array([1.0, 0.0, 1.0])
expect_error("ggl_sgl_batch_step(rho_k = 0)", lib.ggl_sgl_batch_step(h, ptr(rho_bad), ptr(lam3), 0, None, ptr(np.zeros(15))))
rc2, h2 = ctx(3, 10)
expect_error("ggl_snapshot_from(other dimension)", lib.ggl_snapshot_from(h, 0, h2, 0))
expect_error("ggl_snapshot_state_from(k_src out of range)", lib.ggl_snapshot_state_from(h, 0, h, 9))
lib.ggl_ctx_destroy(h2)
# a valid step still works on the ctx that saw all of the above
rc = step(1.0, 0.1, 0.05, 0)
print("valid ggl_admm_step after the probes: rc", rc, last() if rc else "", "norms finite", bool(np.all(np.isfinite(norms[:5]))))
bad += 0 if rc == 0 and np.all(np.isfinite(norms[:5])) else 1
lib.ggl_ctx_destroy(h)

# ---- the stateless operators (the seam solver/admm_solver.py:11 imports) and the batch / snapshot / subset entry points ----
A = np.eye(6)
out6 = np.zeros((1, 6, 6))
b1 = np.ones(1)
expect_error("ggl_phiplus_matrix(K=0)", lib.ggl_phiplus_matrix(0, 6, ptr(b1), ptr(A), ptr(out6), 0))
expect_error("ggl_phiplus_matrix(p=-2)", lib.ggl_phiplus_matrix(1, -2, ptr(b1), ptr(A), ptr(out6), 0))
expect_error("ggl_phiplus_matrix(W=NULL)", lib.ggl_phiplus_matrix(1, 6, ptr(b1), None, ptr(out6), 0))
expect_error("ggl_phiplus_matrix(eig_method=77)", lib.ggl_phiplus_matrix(1, 6, ptr(b1), ptr(A), ptr(out6), 77))
expect_error("ggl_phiplus_matrix(beta=0)", lib.ggl_phiplus_matrix(1, 6, ptr(np.zeros(1)), ptr(A), ptr(out6), 0))
expect_error("ggl_phiplus_matrix(beta=nan)", lib.ggl_phiplus_matrix(1, 6, ptr(np.full(1, np.nan)), ptr(A), ptr(out6), 0))
expect_error("ggl_rank_matrix(eig_method=200)", lib.ggl_rank_matrix(1, 6, ptr(b1), ptr(A), ptr(out6), 200))
expect_error("ggl_phiplus(beta=-1)", lib.ggl_phiplus(1, 6, ptr(-b1), ptr(np.ones(6)), ptr(A), ptr(out6)))
expect_error("ggl_rank_matrix(out=NULL)", lib.ggl_rank_matrix(1, 6, ptr(b1), ptr(A), None, 0))
expect_error("ggl_eigh_batched(A=NULL)", lib.ggl_eigh_batched(1, 6, None, ptr(np.zeros(6)), ptr(out6), 0))
X6 = np.stack([A, A])
expect_error("ggl_prox_p(l1=0)", lib.ggl_prox_p(2, 6, ptr(X6), ctypes.c_double(0.0), ctypes.c_double(0.1), 0, ptr(np.zeros_like(X6))))
expect_error("ggl_prox_p(reg=5)", lib.ggl_prox_p(2, 6, ptr(X6), ctypes.c_double(0.1), ctypes.c_double(0.1), 5, ptr(np.zeros_like(X6))))
expect_error("ggl_prox_p(X=NULL)", lib.ggl_prox_p(2, 6, None, ctypes.c_double(0.1), ctypes.c_double(0.1), 0, ptr(np.zeros_like(X6))))
expect_error("ggl_prox_tv(K=0)", lib.ggl_prox_tv(3, 0, ptr(np.zeros(3)), ctypes.c_double(0.1), ptr(np.zeros(3))))
expect_error("ggl_prox_tv(n=-1)", lib.ggl_prox_tv(-1, 3, ptr(np.zeros(3)), ctypes.c_double(0.1), ptr(np.zeros(3))))
expect_error("ggl_prox_od_1norm(A=NULL)", lib.ggl_prox_od_1norm(6, None, ctypes.c_double(0.1), None, ptr(out6)))
rc, h = ctx(4, 8)
assert rc == 0, last()
S4 = np.stack([np.eye(8)] * 4)
assert lib.ggl_set_S(h, ptr(S4)) == 0 and lib.ggl_set_state(h, ptr(S4), ptr(S4), None, ptr(np.zeros_like(S4))) == 0, last()
r4, l4, o20 = np.ones(4), np.full(4, 0.1), np.zeros(20)
expect_error("ggl_mgl_batch_step(G=3 of K=4)", lib.ggl_mgl_batch_step(h, 3, ptr(r4), ptr(l4), ptr(l4), 0, 0, None, None, ptr(o20)))
expect_error("ggl_mgl_batch_step(G=0)", lib.ggl_mgl_batch_step(h, 0, ptr(r4), ptr(l4), ptr(l4), 0, 0, None, None, ptr(o20)))
sub = ctypes.c_void_p()
expect_error("ggl_ctx_create_subset(idx out of range)", lib.ggl_ctx_create_subset(h, (ctypes.c_int * 2)(1, 9), 2, ctypes.byref(sub)))
expect_error("ggl_ctx_create_subset(m=0)", lib.ggl_ctx_create_subset(h, (ctypes.c_int * 2)(1, 2), 0, ctypes.byref(sub)))
expect_error("ggl_ctx_create_subset(idx=NULL)", lib.ggl_ctx_create_subset(h, None, 2, ctypes.byref(sub)))
expect_error("ggl_get_snapshot_k(k=4 of K=4)", lib.ggl_get_snapshot_k(h, 4, ptr(S4), None))
expect_error("ggl_reset_instance(k=-1)", lib.ggl_reset_instance(h, -1))
expect_error("ggl_failed_reason(k=7)", lib.ggl_failed_reason(h, 7, ptr(np.zeros(2))))
expect_error("ggl_selection_stats(out=NULL)", lib.ggl_selection_stats(h, None))
expect_error("ggl_objective(reg=9)", lib.ggl_objective(h, ctypes.c_double(0.1), ctypes.c_double(0.1), 9, ptr(np.zeros(3))))
expect_error("ggl_trace_start(max_events=-5)", lib.ggl_trace_start(h, -5))
expect_error("ggl_comm_init(nranks=0)", lib.ggl_comm_init(h, 0, 0, ctypes.create_string_buffer(128)))
expect_error("ggl_comm_init(rank >= nranks)", lib.ggl_comm_init(h, 3, 2, ctypes.create_string_buffer(128)))
expect_error("ggl_allreduce_norms without a communicator", lib.ggl_allreduce_norms(h))
rc = lib.ggl_sgl_batch_step(h, ptr(r4), ptr(l4), 0, None, ptr(o20))
print("valid ggl_sgl_batch_step after the probes: rc", rc, last() if rc else "", "sums finite", bool(np.all(np.isfinite(o20))))
bad += 0 if rc == 0 and np.all(np.isfinite(o20)) else 1
lib.ggl_ctx_destroy(h)
print("ok" if bad == 0 else f"{bad} probes misbehaved")
sys.exit(0 if bad == 0 else 1)
