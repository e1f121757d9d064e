"""world_size-2 gloo test of the K-sharded GGL driver (gglasso_amd/dist.py): two CPU processes, each
owning a K-slab, exchange the (p,p) group sums of squares and the 5 residual norms; the result must
equal the single-process solve.  Array work: test-only oracle engine (no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, K, p, out_dir, miss="", latent=False):
    import contextlib
    import io
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gglasso_amd import solver, synth
    from gglasso_amd.dist import ADMM_MGL_sharded, TorchComm, shard_bounds
    from oracle_engine import OracleEngine, SpeculatingOracleEngine
    solver.ENGINE = OracleEngine
    S, _ = synth.make_problem("GGL", K, p, seed=21)
    k0, k1 = shard_bounds(K, world, rank)
    Om0 = np.repeat(np.eye(p)[None], k1 - k0, axis=0)
    comm = TorchComm()
    retries = []
    if miss:
        # the device protocol of the HIP engine (norms all-reduced where they are, speculative Omega-steps validated
        # through a flag on the group-sum all-reduce), emulated on the host with scripted misses
        os.environ["GGL_TEST_MISS"] = miss
        comm.device_norms = True

        class Eng(SpeculatingOracleEngine):
            def close(self):
                retries.append(self.retries)
        solver.ENGINE = Eng
    extra = dict(latent=True, mu1=np.linspace(0.1, 0.2, K)[k0:k1]) if latent else {}
    with contextlib.redirect_stdout(io.StringIO()):
        sol, info = ADMM_MGL_sharded(S[k0:k1], 0.05, 0.02, "GGL", Om0, K, comm, tol=1e-9, rtol=1e-9, measure=True, **extra)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), k0=k0, k1=k1, status=info["status"],
             iters=len(info["residual"]), retries=np.array(retries[-1] if retries else 0), **sol)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,world,miss,latent", [(5, 2, "", False), (4, 2, "", False), (5, 2, "1:2,0:5,1:5,0:9", False),
                                                 (5, 2, "", True), (5, 2, "1:3,0:4", True)])
def test_k_sharded_ggl_equals_single_process(tmp_path, K, world, miss, latent):
    """miss != "": the device protocol of the HIP engine with scripted speculation misses (rank:call) -- rank 1 alone at
    its 2nd speculative step, both ranks at their 5th, rank 0 alone at its 9th.  Every rank must repeat exactly those
    steps (three repeats each), stay in lockstep with the collectives, and end with the unsharded solution.
    latent: the L-step (admm_solver.py:197-205) on every rank's own slab, per-instance mu1, group sums over Omega + L + X."""
    import torch.multiprocessing as mp
    from oracle import ggl_oracle as orc
    from gglasso_amd import synth
    p = 24
    port = _free_port()
    mp.spawn(_worker, args=(world, port, K, p, str(tmp_path), miss, latent), nprocs=world, join=True)
    S, _ = synth.make_problem("GGL", K, p, seed=21)
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    extra = dict(latent=True, mu1=np.linspace(0.1, 0.2, K)) if latent else {}
    ref, rinfo = orc.ADMM_MGL(S, 0.05, 0.02, "GGL", Om0, tol=1e-9, rtol=1e-9, **extra)
    covered = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        k0, k1 = int(z["k0"]), int(z["k1"])
        covered += k1 - k0
        assert str(z["status"]) == rinfo["status"]
        assert int(z["iters"]) == rinfo["iterations"]
        for nm in ("Omega", "Theta", "X") + (("L",) if latent else ()):
            assert np.abs(z[nm] - ref[nm][k0:k1]).max() <= 1e-10, (r, nm)
        if miss:
            # every rank repeats every step at which ANY rank missed (the speculative-call numbers agree across ranks)
            assert int(z["retries"]) == len({t.split(":")[1] for t in miss.split(",")}), (r, int(z["retries"]))
    assert covered == K


def _random_cases(seed, n):
    """n drawn K-sharded problems (same on every rank and in the checker): K, p, lambdas, rho, rho updates, latent, n_samples,
    tolerance or a fixed length, and a script of speculation misses."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        K, p = int(rng.integers(2, 10)), int(rng.choice([3, 8, 16, 17, 24]))
        c = dict(K=K, p=p, seed=int(rng.integers(1 << 30)), lam1=float(10.0 ** rng.uniform(-2, -0.3)),
                 lam2=float(10.0 ** rng.uniform(-2.5, -0.5)), rho=float(10.0 ** rng.uniform(-0.7, 0.7)), upd=bool(rng.random() < 0.6),
                 latent=bool(rng.random() < 0.4), mu=10.0 ** rng.uniform(-0.5, 0.5, K),
                 nk=(rng.integers(5, 300, K) if rng.random() < 0.3 else None))
        if rng.random() < 0.5:
            c.update(tol=1e-20, rtol=1e-20, max_iter=int(rng.integers(1, 14)))
        else:
            c.update(tol=float(10.0 ** rng.uniform(-9, -6)), rtol=float(10.0 ** rng.uniform(-8, -5)), max_iter=400)
        c["miss"] = ",".join(f"{int(rng.integers(2))}:{int(rng.integers(1, 12))}" for _ in range(int(rng.integers(0, 4)))) \
            if rng.random() < 0.5 else ""
        cases.append(c)
    return cases


def _random_worker(rank, world, port, seed, n, out_dir):
    import contextlib
    import io
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gglasso_amd import solver, synth
    from gglasso_amd.dist import ADMM_MGL_sharded, TorchComm, shard_bounds
    from oracle_engine import OracleEngine, SpeculatingOracleEngine
    out = {}
    for i, c in enumerate(_random_cases(seed, n)):
        S, _ = synth.make_problem("GGL", c["K"], c["p"], seed=c["seed"])
        k0, k1 = shard_bounds(c["K"], world, rank)
        comm = TorchComm()
        solver.ENGINE = OracleEngine
        os.environ["GGL_TEST_MISS"] = c["miss"]
        if c["miss"]:
            comm.device_norms = True
            solver.ENGINE = SpeculatingOracleEngine
        extra = dict(latent=True, mu1=c["mu"][k0:k1]) if c["latent"] else {}
        if c["nk"] is not None:
            extra["n_samples"] = c["nk"][k0:k1]
        with contextlib.redirect_stdout(io.StringIO()):
            sol, info = ADMM_MGL_sharded(S[k0:k1], c["lam1"], c["lam2"], "GGL", np.repeat(np.eye(c["p"])[None], k1 - k0, axis=0), c["K"],
                                         comm, tol=c["tol"], rtol=c["rtol"], max_iter=c["max_iter"], rho=c["rho"], update_rho=c["upd"],
                                         measure=True, **extra)
        out[f"{i}_status"] = info["status"]
        out[f"{i}_iters"] = len(info["residual"])
        for nm in sol:
            out[f"{i}_{nm}"] = sol[nm]
    np.savez(os.path.join(out_dir, f"random_rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", [77, 78])
def test_k_sharded_random_cases_equal_the_unsharded_solve(tmp_path, seed):
    """Twenty-four drawn problems (K = 2 .. 9 over two ranks -- uneven slabs, a rank with ONE instance --, p = 3 .. 24, rho updates,
    latent variables with per-instance mu1, n_samples weights, fixed lengths and runs to a tolerance, scripted speculation misses on
    either rank) through the real driver loop on two gloo ranks: every rank's slab equals the oracle's unsharded ADMM_MGL
    (solver/admm_solver.py:13-313), same status, same iteration count."""
    import torch.multiprocessing as mp
    from oracle import ggl_oracle as orc
    from gglasso_amd import synth
    from gglasso_amd.dist import shard_bounds
    n, world = 24, 2
    port = _free_port()
    mp.spawn(_random_worker, args=(world, port, seed, n, str(tmp_path)), nprocs=world, join=True)
    z = [np.load(os.path.join(str(tmp_path), f"random_rank{r}.npz")) for r in range(world)]
    for i, c in enumerate(_random_cases(seed, n)):
        S, _ = synth.make_problem("GGL", c["K"], c["p"], seed=c["seed"])
        extra = dict(latent=True, mu1=c["mu"]) if c["latent"] else {}
        if c["nk"] is not None:
            extra["n_samples"] = c["nk"]
        ref, rinfo = orc.ADMM_MGL(S, c["lam1"], c["lam2"], "GGL", np.repeat(np.eye(c["p"])[None], c["K"], axis=0), tol=c["tol"],
                                  rtol=c["rtol"], max_iter=c["max_iter"], rho=c["rho"], update_rho=c["upd"], **extra)
        for r in range(world):
            k0, k1 = shard_bounds(c["K"], world, r)
            assert int(z[r][f"{i}_iters"]) == rinfo["iterations"], (i, c, r)
            if c["tol"] > 1e-19:          # (at tol = 1e-20 the closing status test passes only for an exactly-zero residual)
                assert str(z[r][f"{i}_status"]) == rinfo["status"], (i, c, r)
            for nm in ("Omega", "Theta", "X") + (("L",) if c["latent"] else ()):
                assert np.abs(z[r][f"{i}_{nm}"] - ref[nm][k0:k1]).max() <= 1e-9 * max(1.0, np.abs(ref[nm]).max()), (i, c, r, nm)


def test_shard_helpers():
    from gglasso_amd.dist import shard_bounds, shard_grid
    for K in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(K, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == K
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    pts = sorted(sum((shard_grid(20, 8, r) for r in range(8)), []))
    assert pts == list(range(20))


def _path_worker(rank, world, port, out_dir):
    import contextlib
    import io
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gglasso_amd import solver, synth
    from gglasso_amd.dist import lambda_path_sharded
    from oracle_engine import OracleEngine
    solver.ENGINE = OracleEngine
    S, _ = synth.make_problem("GGL", 1, 18, N=60, seed=3)
    lams = np.logspace(-0.3, -1.5, 5)
    with contextlib.redirect_stdout(io.StringIO()):
        res = lambda_path_sharded(S[0], lams, tol=1e-9, rtol=1e-9)
    np.savez(os.path.join(out_dir, f"path{rank}.npz"), thetas=np.stack([r[0]["Theta"] for r in res]),
             iters=np.array([r[1]["iterations"] for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_lambda_path_sharded_over_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    from oracle import ggl_oracle as orc
    from gglasso_amd import synth
    port = _free_port()
    mp.spawn(_path_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    S, _ = synth.make_problem("GGL", 1, 18, N=60, seed=3)
    lams = np.logspace(-0.3, -1.5, 5)
    a = np.load(os.path.join(str(tmp_path), "path0.npz"))
    b = np.load(os.path.join(str(tmp_path), "path1.npz"))
    assert np.array_equal(a["thetas"], b["thetas"])          # every rank holds the whole path
    for i, lam in enumerate(lams):
        ref, rinfo = orc.ADMM_SGL(S[0], lam, np.eye(18), tol=1e-9, rtol=1e-9)
        assert int(a["iters"][i]) == rinfo["iterations"]
        assert np.abs(a["thetas"][i] - ref["Theta"]).max() <= 1e-10


def _grid_worker(rank, world, port, out_dir):
    import contextlib
    import io
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gglasso_amd import solver
    from oracle_engine import OracleEngine
    from conftest import load_golden
    from grid_checks import check_mgl_grid_search
    solver.ENGINE = OracleEngine
    solved = []

    class Counting(OracleEngine):
        def __init__(self, S, *a, **k):
            super().__init__(S, *a, **k)
            solved.append(self.K)
    solver.ENGINE = Counting
    with contextlib.redirect_stdout(io.StringIO()):
        # the 3 x 2 grid (6 points) dealt over two ranks: every rank must end with the reference's full tables
        check_mgl_grid_search(load_golden, group=dist.group.WORLD, tags=("GGL_plain", "GGL_latent"))
    np.savez(os.path.join(out_dir, f"grid{rank}.npz"), batch_instances=np.array(solved[:2]))
    dist.barrier()
    dist.destroy_process_group()


def test_mgl_grid_search_sharded_over_two_ranks(tmp_path):
    """The model-selection grid axis over ranks (independent units, replicas only): 6 grid points x K=3
    instances, 3 points per rank as one batch each, results gathered on every rank; tables and selection equal the
    reference's (fixture G16)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_grid_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), f"grid{r}.npz"))
        assert list(d["batch_instances"]) == [9, 9]           # each rank solved 3 of the 6 points (x K = 3) per search
