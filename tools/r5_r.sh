#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_omega_lds.py tests/test_gpu_admm.py tests/test_gpu_batch_isolation.py tests/test_gpu_selection.py tests/test_gpu_latent_rank.py -x -q > $O/pytest_r.txt 2>&1
tail -3 $O/pytest_r.txt
python - <<PY
import time, numpy as np, contextlib, io
from gglasso_amd import synth
from gglasso_amd.batch import ADMM_MGL_batch
for K,p,G in ((5,50,9),(5,40,16),(10,64,6)):
    S,_=synth.make_problem("GGL",K,p,N=2*p,seed=4)
    l1=np.tile(np.logspace(-0.5,-1.5,G),1); l2=np.full(G,0.02)
    ADMM_MGL_batch(S,l1[:2],l2[:2],"GGL",max_iter=3)
    ts=[]
    for rep in range(4):
        t0=time.perf_counter(); res=ADMM_MGL_batch(S,l1,l2,"GGL",tol=1e-7,rtol=1e-6); ts.append(time.perf_counter()-t0)
    its=max(i['iterations'] for _,i in res)
    print(f"MGL grid G={G} K={K} p={p}: {min(ts)*1e3:.2f} ms, {its} batch iterations, {min(ts)/its*1e6:.1f} us per iteration")
PY
