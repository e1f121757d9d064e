"""Worker of tests/test_gpu_multirank.py: one rank of a K-sharded GGL solve over REAL RCCL (one process per GPU, started
by torch.distributed.run).  Every rank also solves the whole problem unsharded on its own GPU and compares its slab.

    python -m torch.distributed.run --nproc-per-node=2 ... tests/rccl_worker.py <out_dir> <K> <p> <comm: capi|torch> <miss_rank>
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, K, p, kind, miss_rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    import torch
    import torch.distributed as dist
    # deliberately NO torch.cuda.set_device: the communicators must carry the device themselves
    dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    from gglasso_amd import solver, synth
    from gglasso_amd.dist import ADMM_MGL_sharded, RcclComm, TorchComm, shard_bounds
    S, _ = synth.make_problem("GGL", K, p, N=2 * p, seed=77)
    k0, k1 = shard_bounds(K, world, rank)
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    comm = RcclComm(device=local) if kind == "capi" else TorchComm(device=f"cuda:{local}")
    seen = []
    real_close = solver.HipEngine.close

    def closing(self):
        if getattr(self, "h", None):
            st = self.ns_stats()
            if getattr(self.lib, "ggl_comm_count", None) is not None and kind == "capi" and self.K == k1 - k0:
                try:
                    st["n_ranks_seen"] = self.comm_count()
                except Exception:  # noqa: BLE001  (the unsharded engine has no communicator)
                    pass
            seen.append(st)
        real_close(self)

    solver.HipEngine.close = closing
    # forced speculation misses on ONE rank: its bounds are deflated, every speculative step fails validation there, and
    # the all-reduced flag must make BOTH ranks repeat the step
    opts = {"spec_factor": 0.9} if rank == miss_rank else {}
    kw = dict(tol=1e-9, rtol=1e-9, max_iter=60)
    with contextlib.redirect_stdout(io.StringIO()):
        sol, info = ADMM_MGL_sharded(S[k0:k1], 0.05, 0.02, "GGL", Om0[k0:k1], K, comm, device=local,
                                     engine_kwargs={"options": opts}, measure=True, **kw)
        st_sharded = seen[-1]
        ref, rinfo = solver.ADMM_MGL(S, 0.05, 0.02, "GGL", Om0, measure=True, **kw)
    err = max(float(np.abs(sol[nm] - ref[nm][k0:k1]).max()) for nm in ("Omega", "Theta", "X"))
    res = {"rank": rank, "world": world, "k0": k0, "k1": k1, "err": err, "status": info["status"],
           "ref_status": rinfo["status"], "iters": len(info["residual"]), "ref_iters": len(rinfo["residual"]),
           "spec_calls": st_sharded["spec_calls"], "spec_misses": st_sharded["spec_misses"],
           "last_parts": st_sharded["last_parts"], "n_ranks_seen": st_sharded.get("n_ranks_seen", world)}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
