"""Per-call host timing of the K-sharded GGL step through torch.distributed/RCCL with ONE rank (dev tool):
step_omega, group_partial, all-reduce of the (p,p) sums, step_finish with device-side vs host-side norms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kk, vv in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29512")):
    os.environ.setdefault(kk, vv)
import numpy as np, torch, torch.distributed as dist
from gglasso_amd import synth, solver, _lib
from gglasso_amd.dist import TorchComm
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
comm = TorchComm(device="cuda:0")
K, p = 4, 500
S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=3)
Om0 = np.stack([np.eye(p)] * K)
eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S), stream=comm.stream_handle, device=0)
nk = np.ones(K)
T = {k: 0.0 for k in ("omega", "gp", "ar_gs", "finish", "ar_n", "finish_sync", "ar_n_host")}
def tick(name, t0):
    t = time.perf_counter(); T[name] += t - t0; return t
for it in range(60):
    if it == 10:
        for k in T: T[k] = 0.0
    t = time.perf_counter()
    eng.step_omega(1.0, False, nk); t = tick("omega", t)
    eng.step_group_partial(1.0, 0.05); t = tick("gp", t)
    comm.allreduce_groupsq(eng); t = tick("ar_gs", t)
    if it % 2 == 0:
        eng.step_finish(1.0, 0.05, 0.01, 'GGL', False, None, 1, defer_norms=True); t = tick("finish", t)
        sq = comm.allreduce_norms_device(eng); t = tick("ar_n", t)
    else:
        sq = eng.step_finish(1.0, 0.05, 0.01, 'GGL', False, None, 1); t = tick("finish_sync", t)
        sq = comm.allreduce_norms(sq); t = tick("ar_n_host", t)
print({k: round(v / 25 * 1e6, 1) for k, v in T.items()})
eng.close()
dist.destroy_process_group()
