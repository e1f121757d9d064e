"""Multi-rank runs on REAL RCCL: with two GPUs visible, two child processes run the
K-sharded GGL driver through ``RcclComm`` (RCCL behind the C ABI, ggl_admm_step_sharded) and through ``TorchComm``
(torch.distributed collectives) -- two-part slabs (K=8, p=500) and uneven slabs (K=5, p=150) with forced speculation
misses on one rank -- against the unsharded solve at 1e-10; and ``bench.py --gpus 2`` started bare.  With one GPU those
are skipped and only the single-rank RCCL start of the bench (GGL_BENCH_FORCE_DIST=1) runs.
The child processes are started as ``python -m torch.distributed.run`` / ``python bench.py`` subprocesses: nothing in this
process execs after it has touched the GPU.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _n_gpus():
    import torch
    return torch.cuda.device_count()      # does not initialise the GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.parametrize("K,p,kind,miss_rank", [(8, 500, "capi", -1), (5, 150, "capi", 1), (5, 150, "torch", 1),
                                                (8, 500, "torch", -1)])
def test_two_rank_rccl_sharded_solve(tmp_path, K, p, kind, miss_rank):
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_worker.py"), str(tmp_path), str(K),
           str(p), kind, str(miss_rank)]
    out = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    res = [json.load(open(os.path.join(str(tmp_path), f"rank{r}.json"))) for r in range(2)]
    assert sum(r["k1"] - r["k0"] for r in res) == K
    for r in res:
        assert r["err"] <= 1e-10, r
        assert r["status"] == r["ref_status"] and r["iters"] == r["ref_iters"], r
        assert r["n_ranks_seen"] == 2, r
        assert r["spec_calls"] >= 1, r
    if miss_rank >= 0:
        # the rank with deflated bounds misses; the other one repeated the same steps although its own bounds held
        assert res[miss_rank]["spec_misses"] >= 1, res
    if (K, p) == (8, 500):
        assert all(r["last_parts"] == 1 for r in res), res      # K=4 slabs: one launch sequence each


def _bench(args, env_extra=None, timeout=900):
    env = _env()
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_bare_start_on_two_gpus():
    """``python bench.py --gpus 2`` with no launcher and no WORLD_SIZE: starts its own two ranks."""
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    line = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--regions", "2"])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["scaling"] == "strong", line
    assert line["roofline"]["frac"] > 0 and "K-slabs of 16" in line["config"]["sharding"], line


def test_bench_single_rank_rccl_start():
    """GGL_BENCH_FORCE_DIST=1: the sharded bench path (RCCL communicator, ggl_admm_step_sharded) with ONE rank."""
    line = _bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--regions", "2", "--no-cpu-baseline"],
                  {"GGL_BENCH_FORCE_DIST": "1"})
    assert line["n_gpus"] == 1 and line["n_ranks_seen"] == 1, line
    assert line["value"] > 0 and line["config"]["options"]["ns_tol"] == 2e-12, line
