import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gglasso_amd import _lib
from gglasso_amd._lib import ptr
lib = _lib.load_dev()      # libggl_hip_dev.so: python -m gglasso_amd.build --dev
pk = np.zeros(1); _lib.check(lib.ggl_dev_mfma_f64_peak(ptr(pk))); print('FP64 MFMA probe peak: %.1f TF/s' % pk[0], flush=True)
for (K, p) in ((32, 500), (64, 500), (32, 1000), (4, 500)):
    for v in (0, 16):
        ms = np.zeros(1)
        _lib.check(lib.ggl_dev_symm_bench(K, p, v, 30, ptr(ms)))
        print(f"K={K:3d} p={p:4d} v{v}: {ms[0]:7.4f} ms  {K*p**3/(ms[0]*1e-3)/1e12:6.2f} TF/s algorithmic; ideal MFMA time at 77 TF/s: {((p+63)//64)*(((p+63)//64)+1)//2*K*2*64*64*((p+63)//64*64)/77e12*1e3:.4f} ms", flush=True)
