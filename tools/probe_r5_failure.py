#!/usr/bin/env python3
"""Probes for the two readings of round 5's intermittent RANK table, development library:
  (A) the runtime's fill + device-to-device copies on one stream: do slices get lost?   ggl_dev_fill_copy_probe
  (B) rocSOLVER's batched dsyevd at the failing shape (K = 6, p = 500) and two others, repeated on the SAME input through the
      stateless operator entry point: is every repetition the first one, bit for bit, and do the eigenvalues add up to the trace?
python tools/probe_r5_failure.py [repetitions]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib, synth
from gglasso_amd._lib import ptr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = _lib.load_dev()
out = (ctypes.c_longlong * 2)()
for (K, p, slices, gap) in ((6, 500, 4, 0), (6, 500, 6, 200), (20, 1000, 5, 0), (64, 100, 9, 0), (6, 501, 4, 50)):
    t0 = time.perf_counter()
    _lib.check(dev.ggl_dev_fill_copy_probe(reps, K, p, slices, gap, out))
    print(f"(A) fill + {slices} device-to-device copies into a fresh ({K},{p},{p}) stack, {gap} us of kernel between them: "
          f"{out[1]} of {out[0] * slices} slices lost in {out[0]} repetitions  ({time.perf_counter() - t0:.1f} s)", flush=True)

lib = _lib.load()
rng = np.random.default_rng(7)
for (K, p) in ((6, 500), (20, 200), (4, 1000)):
    S, _ = synth.make_problem("GGL", K, p, seed=3)
    A = np.ascontiguousarray(S - 0.3 * np.eye(p)[None])                       # indefinite, like the L-step's C - mu I
    D0, Q0 = np.empty((K, p)), np.empty((K, p, p))
    _lib.check(lib.ggl_eigh_batched(K, p, ptr(A), ptr(D0), ptr(Q0), _lib.EIG_ROCSOLVER))
    tr = np.trace(A, axis1=1, axis2=2)
    assert np.abs(D0.sum(axis=1) - tr).max() <= 1e-9 * np.abs(D0).sum(axis=1).max()
    ref = np.linalg.eigvalsh(A)
    assert np.abs(D0 - ref).max() <= 1e-10 * np.abs(ref).max()
    n = max(50, reps // (10 if p >= 500 else 4))
    bad_bits = bad_trace = 0
    t0 = time.perf_counter()
    D, Q = np.empty_like(D0), np.empty_like(Q0)
    for r in range(n):
        _lib.check(lib.ggl_eigh_batched(K, p, ptr(A), ptr(D), ptr(Q), _lib.EIG_ROCSOLVER))
        if not np.array_equal(D, D0):
            bad_bits += 1
            if not np.abs(D.sum(axis=1) - tr).max() <= 1e-8 * np.abs(D0).sum(axis=1).max():
                bad_trace += 1
                print(f"   repetition {r}: eigenvalues off by {np.abs(D - D0).max():.3e}", flush=True)
    print(f"(B) rocsolver_dsyevd_strided_batched K = {K}, p = {p}: {bad_bits} of {n} repetitions differ from the first in any bit, "
          f"{bad_trace} violate the trace identity  ({time.perf_counter() - t0:.1f} s)", flush=True)
