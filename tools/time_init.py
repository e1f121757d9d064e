#!/usr/bin/env python3
"""Where HipEngine.__init__ spends its time when a batch driver is called repeatedly (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import solver, _lib
from gglasso_amd._lib import ptr, check
lib = _lib.load()
import ctypes
for K, p in ((100, 64), (20, 50)):
    S = np.eye(p) + 0.01
    for rep in range(5):
        T = {}
        t0 = time.perf_counter()
        h = _lib._vp()
        check(lib.ggl_ctx_create(0, K, p, 0, None, h)); t1 = time.perf_counter(); T["create"] = t1 - t0
        check(lib.ggl_ctx_set_option(h, _lib.OPTIONS["isolate"], 1.0)); t2 = time.perf_counter(); T["option"] = t2 - t1
        check(lib.ggl_set_S_ex(h, ptr(S), 1)); t3 = time.perf_counter(); T["set_S"] = t3 - t2
        per = (ctypes.c_int * 4)(1, 1, 0, 1)
        Z = np.zeros((p, p))
        check(lib.ggl_set_state_ex(h, ptr(S), ptr(S), None, ptr(Z), per)); t4 = time.perf_counter(); T["set_state"] = t4 - t3
        rho = np.ones(K); lam = np.full(K, 0.1); out = np.zeros((K, 5))
        for _ in range(3):
            check(lib.ggl_sgl_batch_step(h, ptr(rho), ptr(lam), 0, None, ptr(out)))
        t5 = time.perf_counter(); T["3 steps"] = t5 - t4
        check(lib.ggl_snapshot_state_from(h, 0, h, 0)); t6 = time.perf_counter(); T["snapshot"] = t6 - t5
        check(lib.ggl_ctx_destroy(h)); t7 = time.perf_counter(); T["destroy"] = t7 - t6
        print(K, p, rep, {k: round(v * 1e3, 2) for k, v in T.items()})
