#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<PY
import time, numpy as np
from gglasso_amd import synth, solver
from gglasso_amd.batch import ADMM_MGL_batch, ADMM_SGL_batch
def run(fn):
    ts=[]
    for rep in range(5):
        t0=time.perf_counter(); res=fn(); ts.append(time.perf_counter()-t0)
    return min(ts), max(i['iterations'] for _,i in res)
for K,p,G in ((5,50,9),(5,40,16),(10,64,6)):
    S,_=synth.make_problem("GGL",K,p,N=2*p,seed=4)
    l1=np.logspace(-0.5,-1.5,G); l2=np.full(G,0.02)
    for rep in range(2):
        for pinned in (1,0):
            solver.ENGINE_OPTIONS["lds_pinned"]=pinned
            t,its=run(lambda: ADMM_MGL_batch(S,l1,l2,"GGL",tol=1e-7,rtol=1e-6))
            print(f"MGL grid G={G} K={K} p={p} lds_pinned={pinned}: {t*1e3:.2f} ms, {its} batch iterations, {t/its*1e6:.1f} us per iteration")
S,_=synth.make_problem("GGL",1,50,N=100,seed=1235)
lam=np.logspace(0,-2,20)
for rep in range(2):
    for pinned in (1,0):
        solver.ENGINE_OPTIONS["lds_pinned"]=pinned
        t,its=run(lambda: ADMM_SGL_batch(S[0],lam,Omega_0=np.eye(50),X_0=np.eye(50),tol=1e-7,rtol=1e-7))
        print(f"SGL grid 20 x p=50 lds_pinned={pinned}: {t*1e3:.2f} ms, {its} batch iterations")
PY
for pinned in 1 0 1 0; do
python bench.py --workload ggl_K256_p64 --steps 30 --warmup 8 --regions 5 --no-cpu-baseline --opt lds_pinned=$pinned 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ggl_K256_p64 lds_pinned=$pinned', round(d['value'],1), d['ms_per_step'])"
done
