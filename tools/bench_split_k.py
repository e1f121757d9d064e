"""Round 3 (the small-K regime): k_symm_sk -- 32x32 tiles whose k-range is split over the four waves of
a workgroup (each wave a whole 2x2-block tile over its own slabs, own DMA pipeline, no barrier in the main loop; variants
41-45) and additionally over 2 / 4 workgroups with a memory-side reduction (46-50) -- against the shipped 32x32
direct-to-LDS kernel (variant 20).  Checks every variant against NumPy, then times them.  Dev build.
-> profiles/r3_small_batch_split_k.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()
rng = np.random.default_rng(0)
for (K, p) in ((4, 500), (2, 500), (8, 500), (20, 200), (1, 130), (16, 500), (3, 200), (12, 96)):
    A = rng.standard_normal((K, p, p)); A = 0.5 * (A + A.transpose(0, 2, 1))
    E = rng.standard_normal((K, p, p)); E = 0.5 * (E + E.transpose(0, 2, 1))
    coef = rng.uniform(0.5, 1.5, (K, 5))
    ref = coef[:, 0, None, None] * np.eye(p)[None] + coef[:, 1, None, None] * (A @ A) + coef[:, 2, None, None] * E
    ref2 = coef[:, 3, None, None] * np.eye(p)[None] + coef[:, 4, None, None] * ref
    for v in (41, 46, 47, 48, 49, 50):
        if K in (3, 12) and v >= 46:
            continue
        C, C2 = np.empty_like(A), np.empty_like(A)
        for rep in range(2):
            _lib.check(lib.ggl_dev_symm(K, p, ptr(A), ptr(A), ptr(E), ptr(np.ascontiguousarray(coef)), ptr(C), ptr(C2), v))
        err = np.abs(C - ref).max() / np.abs(ref).max()
        err2 = np.abs(C2 - ref2).max() / np.abs(ref2).max()
        print(K, p, v, "err %.1e %.1e sym %s" % (err, err2, np.array_equal(C, C.transpose(0, 2, 1))), flush=True)
def t(K, p, v):
    ms = np.zeros(1); _lib.check(lib.ggl_dev_symm_bench(K, p, v, 40, ptr(ms))); return ms[0] * 1e3
print("us per launch: variants 20 41 44 | split 2: 46 48 | split 4: 47 49 50")
for (K, p) in ((2, 500), (4, 500), (8, 500), (16, 500), (20, 200), (4, 1000), (1, 500)):
    print(K, p, " ".join("%6.1f" % t(K, p, v) for v in (20, 41, 44, 46, 48, 47, 49, 50)), flush=True)
