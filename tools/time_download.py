#!/usr/bin/env python3
"""ggl_get_state of a (K,p,p) problem with 1 .. 16 page-touching threads (GGL_OPT_DOWNLOAD_THREADS): milliseconds, GB/s, and that
the bytes are the same.   python tools/time_download.py [K p]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import solver, synth  # noqa: E402

for K, p in ((32, 500), (20, 1000), (50, 500)) if len(sys.argv) < 3 else ((int(sys.argv[1]), int(sys.argv[2])),):
    S, _ = synth.make_problem("GGL", K=K, p=p, N=2 * p, seed=3)
    Om0 = np.stack([np.eye(p)] * K)
    eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S))
    for _ in range(3):
        eng.step(1.0, 0.05, 0.01, "GGL", False, None, np.ones(K))
    ref = eng.state()
    from gglasso_amd._lib import check, ptr
    # (a) into fresh arrays every time (what a solve's return does: first touch of the pages is part of it); (b) into arrays that
    # exist and have been written before
    warm = {k: np.zeros_like(v) for k, v in ref.items()}
    for nt in (1, 8, 1, 4, 8, 16, 1, 8):
        eng.set_option("download_threads", nt)
        tf, tw = [], []
        for _ in range(4):
            t0 = time.perf_counter()
            st = eng.state()
            tf.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            check(eng.lib.ggl_get_state(eng.h, ptr(warm['Omega']), ptr(warm['Theta']), ptr(warm['L']), ptr(warm['X'])))
            tw.append(time.perf_counter() - t0)
        nbytes = sum(a.nbytes for a in st.values())
        same = all(np.array_equal(ref[k], st[k]) and np.array_equal(ref[k], warm[k]) for k in ref)
        print(f"K = {K}, p = {p}: {nt} thread(s)  fresh arrays {min(tf) * 1e3:7.2f} ms {nbytes / min(tf) / 1e9:6.1f} GB/s (median "
              f"{sorted(tf)[len(tf) // 2] * 1e3:.2f})   touched arrays {min(tw) * 1e3:7.2f} ms {nbytes / min(tw) / 1e9:6.1f} GB/s (median "
              f"{sorted(tw)[len(tw) // 2] * 1e3:.2f})  same bytes: {same}", flush=True)
    eng.close()
