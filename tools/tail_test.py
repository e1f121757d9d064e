import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load()
pk = np.zeros(1); _lib.check(lib.ggl_dev_mfma_f64_peak(ptr(pk))); print('FP64 MFMA probe peak: %.1f TF/s' % pk[0], flush=True)
for (K, p) in ((32, 500), (28, 500), (32, 1000)):
    for v in (0, 1, 4):
        ms = np.zeros(1)
        _lib.check(lib.ggl_dev_symm_bench(K, p, v, 30, ptr(ms)))
        T = (p + 63) // 64
        wgs = T * (T + 1) // 2 * K
        mf = wgs * 2 * 64 * 64 * (T * 64) / (ms[0] * 1e-3) / 1e12
        print(f"K={K:3d} p={p:4d} v{v}: {ms[0]:7.4f} ms  WGs={wgs:5d}  MFMA rate {mf:5.1f} TF/s ({mf/78.6*100:4.1f}% of peak)", flush=True)
