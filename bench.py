#!/usr/bin/env python3
"""Benchmark of the MI355X ADMM hot path: ADMM iterations/second on the Group Graphical Lasso
workload BASELINE.json quotes the metric on, (K=32, p=500), fp64.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full ADMM iteration (Omega-step: W formation + matrix-function products; Theta-step:
group prox over the (K,p,p) stack; dual update; the five stopping-test norms; rho rule on the host)
driven by the same host loop ``ADMM_MGL`` uses (gglasso_amd/solver.py::_run_admm) with tol=rtol=1e-20
so that it cannot exit early.  S, Omega, Theta, X are resident in HBM before the timed region starts.

The timed region is EXACTLY ``steps`` iterations between barrier + synchronize on both sides; it is run
``--regions`` times back to back (default 7; 20 iterations are only ~20 ms of wall clock) and ``value`` /
``ms_per_step`` are those of the MEDIAN region, with the fastest and slowest one reported next to it.

N > 1, one rank per GPU: the K=32 stack is sharded into K/N slabs (strong scaling); per iteration one (p,p) fp64
all-reduce of the group sums of squares and one 5-scalar all-reduce of the residual norms go over RCCL.  Started either
by a launcher (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK / WORLD_SIZE in the
environment) or bare (``python bench.py --gpus N``): the bare process then starts the N ranks ITSELF as a child
``torch.distributed.run`` on 127.0.0.1 before anything in it has touched the GPU (it never does), relays rank 0's JSON
line and exits with the child's code.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the phase that dominates the iteration, timed live with HIP events on the ctx stream
  cpu_baseline -- the CPU oracle (NumPy/LAPACK + C prox) on this box's host cores, bounded sample
"""
import argparse
import contextlib
import io
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_MFMA_PEAK_TF = 78.6     # AMD MI355X datasheet, FP64 matrix (= FP64 vector); the guide lists no fp64 row

WORKLOADS = {
    # name: (reg, K, p, latent, lambda1, lambda2, seed)
    "ggl_K32_p500": ("GGL", 32, 500, False, 0.05, 0.01, 1239),
    "ggl_K20_p200": ("GGL", 20, 200, False, 0.05, 0.01, 1236),
    "fgl_K50_p500_latent": ("FGL", 50, 500, True, 0.05, 0.01, 1237),
    "ggl_K256_p1000": ("GGL", 256, 1000, False, 0.05, 0.01, 1238),
    "ggl_K16_p500": ("GGL", 16, 500, False, 0.05, 0.01, 1239),     # per-GPU slabs of the headline at 2 / 4 / 8 GPUs
    "ggl_K8_p500": ("GGL", 8, 500, False, 0.05, 0.01, 1239),
    "ggl_K4_p500": ("GGL", 4, 500, False, 0.05, 0.01, 1239),
    "ggl_K32_p1000": ("GGL", 32, 1000, False, 0.05, 0.01, 1238),   # per-GPU slab of C5 at 8 GPUs
    # p <= 128: the one-workgroup-per-matrix LDS Jacobi eigensolver (csrc/eig_jacobi.hip), the north_star's small-p kernel
    "ggl_K64_p100": ("GGL", 64, 100, False, 0.05, 0.01, 1241),
    "ggl_K256_p64": ("GGL", 256, 64, False, 0.05, 0.01, 1242),
    "ggl_K32_p128": ("GGL", 32, 128, False, 0.05, 0.01, 1243),
    # odd p next to its even neighbour (the direct-to-LDS product kernel needs 16-byte rows)
    "ggl_K32_p501": ("GGL", 32, 501, False, 0.05, 0.01, 1239),
    "ggl_K32_p502": ("GGL", 32, 502, False, 0.05, 0.01, 1239),
    "ggl_K20_p201": ("GGL", 20, 201, False, 0.05, 0.01, 1236),
    "ggl_K20_p202": ("GGL", 20, 202, False, 0.05, 0.01, 1236),
    # C2 of BASELINE.json: Single GL p = 1000 over a 20-point lambda1 grid as ONE batch of 20 independent problems
    # (gglasso_amd.batch / ggl_sgl_batch_step; lambda1 = logspace(0, -2, 20), own rho per point).  A step = one batched
    # iteration of all 20 points; lambda1 / lambda2 of the tuple are unused.
    "sgl_p1000_grid20": ("SGL", 20, 1000, False, 0.0, 0.0, 1235),
}

# product-kernel instances of csrc/gemm_sym.hip by the variant number ggl_ns_stats reports
VARIANT_NAMES = {0: "k_symm_tn<64,16,32,32> (register-staged)", 9: "k_symm_tn<32,32,16,16> (register-staged)",
                 40: "k_omega_chain<16,3,64> (persistent product chain, per-instance dependencies; 64x64 direct-to-LDS tiles)",
                 16: "k_symm_dl<16,2,0,64> (direct-to-LDS)", 17: "k_symm_dl<16,3,0,64> (direct-to-LDS)",
                 20: "k_symm_dl<32,2,0,32> (direct-to-LDS, 32x32 tiles)",
                 41: "k_omega_lds<PT> (p <= 64: the whole Omega-step in one launch, one workgroup per instance, chain resident in LDS)"}


def phase_model(phase, reg, K, p, latent, eig_jacobi, omega_ns=False):
    """(bound, algorithmic amount per launch, unit) of one phase.  B = one fp64 stack = 8*K*p^2 bytes.
    Bytes follow SURVEY.md section 8(d) (compulsory stack passes); flops are LAPACK-equivalent
    9 p^3 per eigendecomposition and 2 p^3 per reconstruction."""
    B = 8.0 * K * p * p
    if phase == "form_W":
        return "hbm", (5 if latent else 4) * B, "GB/s"
    if phase == "eig_omega" and omega_ns:
        # one launch = one product of two commuting symmetric matrices per instance: only the upper
        # triangle is needed, p^3 flop (of the 2 p^3 of a general product)
        return "mfma", 1.0 * K * p ** 3, "TFLOP/s"
    if phase in ("eig_omega", "eig_L"):
        fl = (11.0 if eig_jacobi else 9.0) * K * p ** 3
        return "mfma", fl, "TFLOP/s"
    if phase in ("recon_omega", "recon_L"):
        return "mfma", 2.0 * K * p ** 3, "TFLOP/s"
    if phase == "theta":
        return "hbm", (4 if latent else 5) * B, "GB/s"    # latent: Omega,L,X -> Theta ; else Omega,X,Omega_prev -> Theta,X
    if phase == "dual":
        return "hbm", 6 * B, "GB/s"
    return "hbm", 0.0, "GB/s"


def committed_profile(kernel_key):
    """What the committed rocprofv3 runs of this same command say about the dominant kernel -- NOT measured by this
    run: HBM bytes per launch from the two PMC passes (tools/profile_round.sh -> tools/summarize_pmc.py, FETCH_SIZE
    doubled for the 16-byte-per-lane DMA kernel as MI355X_MICROARCH.md prescribes) and the kernel's average duration
    in the --kernel-trace --stats summary.  Returns {} when nothing is committed."""
    import csv
    import glob
    out = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if files:
        data = json.load(open(files[-1]))
        for name, c in data.items():
            if name.startswith(kernel_key) and "hbm_bytes_per_launch" in c:
                out["profiled_traffic_bytes"] = c["hbm_bytes_per_launch"]
                out["profiled_traffic_source"] = "profiles/" + os.path.basename(files[-1])
                break
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_kernel_stats.csv")))
    if files:
        for row in csv.DictReader(open(files[-1])):
            if kernel_key in row["Name"]:
                out["profiled_kernel_avg_ms"] = float(row["AverageNs"]) * 1e-6
                out["profiled_kernel_source"] = "profiles/" + os.path.basename(files[-1])
                break
    return out


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def grid_cpu_baseline(S, lam, threads, eng, run, points=(0, 9, 19), iters=4):
    """CPU oracle on a bounded sample of the lambda grid: `iters` ADMM_SGL iterations of `points` of the grid (the oracle's
    eigh on p = 1000 takes ~1 s per iteration and point), scaled to batched iterations per second (one batched iteration =
    all len(lam) points); and the parity of the engine's batch at those points after the same iterations."""
    from oracle import ggl_oracle as orc
    p = S.shape[0]
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:  # noqa: BLE001
        limiter = contextlib.nullcontext()
    refs = {}
    with limiter:
        t0 = time.perf_counter()
        for k in points:
            refs[k], _ = quiet(orc.ADMM_SGL, S, lam[k], np.eye(p), max_iter=iters, tol=1e-20, rtol=1e-20)
        dt = time.perf_counter() - t0
    per_point_iter = dt / (len(points) * iters)
    cpu = {"value": 1.0 / (per_point_iter * len(lam)), "unit": "ADMM iters/s", "cores": threads, "kind": "port",
           "sample": f"{iters} ADMM_SGL iterations of {len(points)} of the {len(lam)} grid points (oracle/ggl_oracle.py: "
                     f"numpy.linalg.eigh + soft threshold) on {threads} BLAS threads, {dt:.1f} s; scaled to batched iterations "
                     f"(all {len(lam)} points) per second"}
    eng.profile(0)
    eye = np.broadcast_to(np.eye(p), (len(lam), p, p))
    eng.set_state(np.ascontiguousarray(eye), np.ascontiguousarray(eye), np.zeros((len(lam), p, p)))
    run(iters, 1.0)
    worst, fro = 0.0, 0.0
    for k in points:
        got = eng.state_k(k)
        worst = max(worst, max(float(np.abs(got[nm] - refs[k][nm]).max()) for nm in ("Omega", "Theta", "X")))
        fro = max(fro, float(np.linalg.norm(got["Theta"] - refs[k]["Theta"])))
    parity = {"iters": iters, "against": f"oracle ADMM_SGL at grid points {list(points)}, identity start, rho rule on",
              "max_abs_vs_oracle": worst, "theta_fro_vs_oracle": fro, "tolerance": "north_star: Theta within 1e-8 Frobenius"}
    return cpu, parity


def cpu_baseline(S, reg, lambda1, lambda2, latent, mu1, iters, threads):
    """CPU oracle (a port: NumPy eigh + C prox), bounded sample of the same workload, on `threads` BLAS threads
    (LAPACK's eigh at p = 500 gets SLOWER beyond a few threads; all 128 of the box's cores is a strawman)."""
    from oracle import ggl_oracle as orc
    K, p, _ = S.shape
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:  # noqa: BLE001
        limiter = contextlib.nullcontext()
        threads = os.cpu_count() or 1
    with limiter:
        orc.ADMM_MGL(S, lambda1, lambda2, reg, Om0, max_iter=1, tol=1e-20, rtol=1e-20, latent=latent, mu1=mu1)
        t0 = time.perf_counter()
        sol, _ = orc.ADMM_MGL(S, lambda1, lambda2, reg, Om0, max_iter=iters, tol=1e-20, rtol=1e-20, latent=latent, mu1=mu1)
        dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "ADMM iters/s", "cores": int(threads), "kind": "port",
            "sample": f"{iters} ADMM iterations of the same ({reg}, K={K}, p={p}) problem from the identity start "
                      f"(oracle/ggl_oracle.py: numpy.linalg.eigh + C prox_p) on {threads} BLAS threads of the box's "
                      f"{os.cpu_count()} cores, {dt:.1f} s"}, sol


def spawn_command(gpus, argv, port=None):
    """The child launcher a bare ``python bench.py --gpus N`` (N > 1, no WORLD_SIZE in the environment) starts."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(gpus, argv, runner=None):
    """Run the N ranks as a child process group and relay their output.  The calling process has not initialised the
    GPU (nothing here imports torch or loads the HIP library) and does not exec: it waits for the child and returns its
    exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = spawn_command(gpus, argv)
    proc = (runner or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in (proc.stdout or "").splitlines():
        print(line, flush=True)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=7, help="how often the timed region of `steps` iterations is run")
    ap.add_argument("--workload", default="ggl_K32_p500", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-iters", type=int, default=24)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eig", type=int, default=0, help="GGL_EIG_* selector (0 auto)")
    ap.add_argument("--comm", default="capi", choices=["capi", "torch"],
                    help="N > 1: RCCL behind the C ABI (one call per iteration) or torch.distributed collectives")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="ctx option (gglasso_amd._lib.OPTIONS), e.g. --opt pipeline=0; repeatable")
    ap.add_argument("--no-exact-region", action="store_true",
                    help="skip the one extra timed region at ns_tol=0 (value_exact)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare start on a multi-GPU node: become the launcher (before any GPU call; never re-exec)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    from gglasso_amd import synth, solver, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GGL_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (1-GPU dev boxes)
    distributed = world > 1 or bool(os.environ.get("GGL_BENCH_FORCE_DIST"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it bare (it launches its own ranks) "
                         "or under torch.distributed.run with --nproc-per-node equal to --gpus")
    if torch.cuda.device_count() < world:       # device_count() does not initialise the GPU
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    comm = None
    if distributed:
        if world == 1:      # GGL_BENCH_FORCE_DIST=1 without a launcher: single-rank rendezvous on the loopback
            for kk, vv in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29511")):
                os.environ.setdefault(kk, vv)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist
        from gglasso_amd.dist import RcclComm, TorchComm, shard_bounds
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        comm = RcclComm(device=local_rank) if args.comm == "capi" else TorchComm(device=f"cuda:{local_rank}")

    reg, K, p, latent, l1, l2, seed = WORKLOADS[args.workload]
    is_grid = reg == "SGL"
    if is_grid:
        S1, _ = synth.make_problem("GGL", 1, p, N=2 * p, seed=seed)
        S = np.broadcast_to(S1[0], (K, p, p))            # uploaded once, replicated on the device (ggl_set_S_ex)
        lam_grid = np.logspace(0, -2, K)
        assert not distributed, "the lambda grid runs as replicas (gglasso_amd.dist.lambda_path_sharded), not K-sharded"
    else:
        S, _ = synth.make_problem(reg, K, p, N=2 * p, seed=seed)
    mu1 = 0.5 * np.ones(K) if latent else None
    if distributed:
        assert reg == "GGL" and not latent, "only GGL shards across K; other workloads run as replicas"
        k0, k1 = shard_bounds(K, world, rank)
        S_loc = np.ascontiguousarray(S[k0:k1])
    else:
        k0, k1, S_loc = 0, K, S
    Kl = k1 - k0
    Om0 = np.repeat(np.eye(p)[None], Kl, axis=0)
    options = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in args.opt}

    def make_engine():
        stream = comm.stream_handle if distributed else None  # the communicator's dedicated stream (gglasso_amd/dist.py)
        return solver.HipEngine(S_loc, Om0, Om0, np.zeros_like(S_loc), eig=args.eig, device=local_rank, stream=stream,
                                options=options)

    eng = make_engine()
    comm_note = None
    if distributed and args.comm == "capi":
        # RCCL behind the C ABI is a second RCCL instance in this process beside torch's own.  Should it not come up, ALL ranks
        # fall back to torch.distributed's collectives together (and the line says so) rather than lose the run.  Only
        # SYMMETRIC failures can be handled that way -- the library missing its RCCL symbols or rank 0 failing to create
        # the unique id (dist.CommIdError is raised on every rank), the forced failure of the test: every rank then reaches
        # the all-reduce of the flag below.  A rank that fails alone inside ncclCommInitRank leaves the others in RCCL's
        # rendezvous; entering a torch collective there would only add a second hang, so it exits non-zero and
        # the launcher tears the job down.
        from gglasso_amd.dist import CommIdError
        err = None
        try:
            if os.environ.get("GGL_BENCH_FAIL_CAPI"):
                raise CommIdError("GGL_BENCH_FAIL_CAPI is set (test of the fallback)")
            comm.attach(eng)
        except CommIdError as e:
            err = f"{type(e).__name__}: {e}"
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}: RCCL communicator behind the C ABI failed on this rank ({type(e).__name__}: {e}); "
                  "leaving the job", file=sys.stderr, flush=True)
            os._exit(3)
        flag = torch.tensor([0 if err is None else 1], device=f"cuda:{local_rank}")
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if int(flag.item()) != 0:
            comm_note = "RCCL behind the C ABI did not come up on every rank (" + (err or "another rank failed") + \
                        "); fell back to torch.distributed collectives"
            print("bench.py: " + comm_note, file=sys.stderr)
            eng.close()
            args.comm = "torch"
            comm = TorchComm(device=f"cuda:{local_rank}")
            eng = make_engine()
    nk = np.ones(Kl)
    mu_loc = None if mu1 is None else mu1[k0:k1]

    def run(iters, rho):
        if is_grid:
            # the loop of gglasso_amd.batch.ADMM_SGL_batch without the stopping test (fixed iteration count): every point
            # keeps its own rho (single_admm_solver.py:196-206)
            rhos = np.full(K, float(rho)) if np.isscalar(rho) else rho
            dimk = (p * p + p) / 2
            for _ in range(iters):
                sq = eng.sgl_batch_step(rhos, lam_grid, False, None)
                fac = np.ones(K)
                for k in range(K):
                    r_t, s_t, _, _ = solver.residuals_from_norms(sq[k], rhos[k], 1e-20, 1e-20, dimk)
                    rn = solver.next_rho(rhos[k], r_t, s_t)
                    fac[k] = rhos[k] / rn
                    rhos[k] = rn
                if np.any(fac != 1.0):
                    eng.scale_X_batch(fac)
            return rhos
        info, rho = quiet(solver._run_admm, eng, reg, K, p, l1, l2, latent, mu_loc, nk, rho, 1e-20, 1e-20, 'boyd',
                          True, iters, False, False, "Multiple", comm=comm)
        return rho

    def fence():
        if distributed:
            torch.distributed.barrier()
        eng.sync()
        torch.cuda.synchronize()

    # Every timed region is the SAME piece of work: back to the identity start, `warmup` untimed iterations, then exactly
    # `steps` timed ones.  (The solve converges to the 1e-20 tolerances' floor after ~150 iterations and would stop by
    # itself, so the regions cannot simply follow each other.)
    # The start point is restored DEVICE TO DEVICE (a copy kept in HBM): re-uploading it from the host leaves the GPU idle for
    # ~20 ms per region, after which the chip needs ~15 iterations to come back to its loaded clock state -- measured with
    # tools/iter_profile.py: iterations 5..24 take 0.863 ms each after an upload and 0.811 ms after a device-side restore,
    # 0.800 ms in the steady state.  A solve (or a grid of them) keeps the GPU busy; so do the regions now.
    eng.save_state()

    def one_region():
        """-> (seconds of the timed `steps` iterations, max over ranks; ns_stats before; ns_stats after; rho)"""
        eng.profile(0)
        eng.restore_state()
        rho = 1.0
        if args.warmup > 0:
            rho = run(args.warmup, rho)
        # live HIP-event timing of the dominant (eigen / matrix-function) phases only during the timed regions:
        # every extra event pair costs a few microseconds of host time per iteration
        eng.profile(2)
        na = eng.ns_stats()
        fence()
        t0 = time.perf_counter()
        rho = run(args.steps, rho)
        fence()
        dt = time.perf_counter() - t0
        nb = eng.ns_stats()
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, na, nb, rho

    effective_options = {name: eng.get_option(name) for name in _lib.OPTIONS}      # what the library actually runs with
    eng.profile_read(reset=True)
    region_s = []
    ns0 = ns1 = None
    rho = 1.0
    for _ in range(max(1, args.regions)):
        dt, na, nb, rho = one_region()
        if ns0 is None:
            ns0 = na
        ns1 = nb if ns1 is None else {k: (ns1[k] + nb[k] - na[k]) if k not in ("last_parts", "last_variant") else nb[k]
                                      for k in nb}
        region_s.append(dt)
    dt = statistics.median(region_s)
    timed_iters = args.steps * len(region_s)
    prof = eng.profile_read(reset=True)
    pipe_totals = eng.pipeline_stats() if hasattr(eng, "pipeline_stats") else None
    # untimed extra pass with every phase instrumented, for the per-phase breakdown
    eng.profile(1)
    extra_iters = min(10, args.steps)
    run(extra_iters, rho)
    prof_all = eng.profile_read(reset=True)
    for ph, v in prof_all.items():
        if ph not in ("eig_omega", "eig_omega2", "eig_L"):
            prof[ph] = v
    # one more region with the Omega-step iterated to fp64 resolution (GGL_OPT_NS_TOL = 0): the same record then carries
    # the rate at the accuracy of an eigendecomposition next to the rate at the default stopping tolerance
    exact = None
    omega_is_ns = (args.eig == _lib.EIG_NEWTON_SCHULZ) or (args.eig == _lib.EIG_AUTO and p > _lib.NS_MIN_P)
    if omega_is_ns and not args.no_exact_region and effective_options["ns_tol"] > 5e-16:
        eng.set_option("ns_tol", 0.0)
        dte, na, nb, _ = one_region()
        eng.profile(0)
        eng.profile_read(reset=True)
        exact = {"value": args.steps / dte, "ms_per_step": dte / args.steps * 1e3, "ns_tol": 0.0, "regions": 1,
                 "products_per_step": (nb["units"] - na["units"]) / max(1, nb["calls"] - na["calls"])}
        eng.set_option("ns_tol", effective_options["ns_tol"])
    # CPU oracle on this box's host cores (bounded sample) -- and, from the SAME oracle run, the parity of this engine
    # after the same number of iterations of the same problem (outside every timed region)
    cpu = parity = None
    if rank == 0 and not distributed and not args.no_cpu_baseline and is_grid:
        cpu, parity = grid_cpu_baseline(S1[0], lam_grid, args.cpu_threads, eng, run)
    elif rank == 0 and not distributed and not args.no_cpu_baseline:
        cpu, ref = cpu_baseline(S, reg, l1, l2, latent, mu1, args.cpu_iters, args.cpu_threads)
        eng.profile(0)
        eng.set_state(Om0, Om0, np.zeros_like(S_loc))
        run(args.cpu_iters, 1.0)
        got = eng.state()
        names = ("Omega", "Theta", "X") + (("L",) if latent else ())
        parity = {"iters": args.cpu_iters, "against": "oracle/ggl_oracle.py (the cpu_baseline run), identity start, rho rule on",
                  "max_abs_vs_oracle": max(float(np.abs(got[nm] - ref[nm]).max()) for nm in names),
                  "theta_fro_vs_oracle": float(np.linalg.norm(got["Theta"] - ref["Theta"])),
                  "theta_fro": float(np.linalg.norm(ref["Theta"])), "ns_tol": effective_options["ns_tol"],
                  "tolerance": "north_star: Theta within 1e-8 Frobenius of the reference"}
    n_ranks_seen = eng.comm_count() if (distributed and args.comm == "capi") else (world if distributed else 1)
    rank_totals = eng.rank_stats() if latent else None
    group_totals = eng.group_stats() if hasattr(eng, "group_stats") else None
    eng.close()

    if rank == 0:
        eig_jacobi = (args.eig == _lib.EIG_JACOBI) or (args.eig == _lib.EIG_AUTO and p <= _lib.NS_MIN_P)
        phases = {ph: {"ms_per_launch": ms / cnt, "launches": cnt} for ph, (ms, cnt) in prof.items() if cnt}
        if "eig_omega" in phases and ns1["launches"] > ns0["launches"]:
            # the Newton-Schulz Omega-step is timed in one event pair (speculative step: no host sync inside) or two
            # (before / after the spectral-bound sync); either way: total elapsed time / product launches
            tot = sum(phases[q]["ms_per_launch"] * phases[q]["launches"] for q in ("eig_omega", "eig_omega2")
                      if q in phases)
            nl = max(1, ns1["launches"] - ns0["launches"])
            phases["eig_omega"] = {"ms_per_launch": tot / nl, "launches": nl}
            phases.pop("eig_omega2", None)
        hot = ("eig_omega", "eig_L")

        def per_iter(ph):   # hot phases were timed over the timed regions, the others over the extra pass
            return phases[ph]["ms_per_launch"] * phases[ph]["launches"] / (timed_iters if ph in hot else extra_iters)

        dom = max(phases, key=per_iter)
        omega_ns = (args.eig == _lib.EIG_NEWTON_SCHULZ) or (args.eig == _lib.EIG_AUTO and not eig_jacobi)
        bound, amount, unit = phase_model(dom, reg, Kl, p, latent, eig_jacobi, omega_ns)
        # what the library dispatched in its last matrix-function step (ggl_ns_stats), not a copy of its rules
        parts, variant = ns1["last_parts"], ns1["last_variant"]
        ns_kernel = VARIANT_NAMES.get(variant, f"product-kernel variant {variant}")
        kernel_name = {"eig_omega": ns_kernel + (" -- W, bound, schedule and all Newton-Schulz products" if variant == 41 else
                                                 " -- Newton-Schulz product") if omega_ns else
                       ("k_jacobi" if eig_jacobi else "rocsolver_dsyevd (library, many kernels)"),
                       "theta": "k_theta_ggl" if reg == "GGL" else ("k_theta_fgl" if reg == "FGL" else "k_theta_sgl"),
                       "allreduce_groupsq": "ncclAllReduce p(p+1)/2+1 fp64 (packed upper triangle + flag)", "allreduce_norms": "ncclAllReduce 5 fp64",
                       "recon_omega": "k_recon", "recon_L": "k_recon", "form_W": "k_form_W",
                       "dual": "k_dual_update", "eig_L": "k_jacobi" if eig_jacobi else "rocsolver_dsyevd"}.get(dom, dom)
        sec = phases[dom]["ms_per_launch"] * 1e-3
        step_amount = None
        if dom == "eig_L" and omega_ns and ns1["rank_launches"] > ns0["rank_launches"]:
            # L-step by sign Newton-Schulz: the phase is (2n+1) symmetric products of K p^3 flop each
            launches = ns1["rank_launches"] - ns0["rank_launches"]
            kernel_name = ns_kernel + " -- sign Newton-Schulz product, L-step"
            bound, unit = "mfma", "TFLOP/s"
            amount = 1.0 * Kl * p ** 3
            sec = phases[dom]["ms_per_launch"] * phases[dom]["launches"] * 1e-3 / launches
            phases[dom] = {"ms_per_launch": sec * 1e3, "launches": launches}
            step_amount = amount * launches / timed_iters
        if dom == "eig_omega" and omega_ns:
            # the Omega-step's launches differ in size (a pair launch carries two products): average over
            # the step = algorithmic flop of all its launches / their total duration (HIP events)
            units = ns1["units"] - ns0["units"]
            launches = ns1["launches"] - ns0["launches"]
            amount = units * 1.0 * Kl * p ** 3 / launches
            step_amount = units * 1.0 * Kl * p ** 3 / timed_iters
        if bound == "hbm":
            achieved, peak, scale = amount / sec / 1e9, HBM_PEAK_GBS, 1e9
        else:
            achieved, peak, scale = amount / sec / 1e12, FP64_MFMA_PEAK_TF, 1e12
        if step_amount is None:
            step_amount = amount * phases[dom]["launches"] / (timed_iters if dom in hot else extra_iters)
        its = args.steps / dt
        iter_bytes = (120.0 if latent else 72.0) * Kl * p * p          # SURVEY.md 8(d), per GPU
        default_cfg = (args.workload == "ggl_K32_p500" and args.eig == 0 and not args.opt)
        roof = {"kernel": kernel_name, "phase": dom, "launches_per_step": phases[dom]["launches"] / timed_iters,
                "bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                # the same algorithmic amount against the WHOLE iteration's wall time (everything that is not this
                # phase counts as zero work): the figure a reader can check against ms_per_step alone
                "frac_of_step": step_amount / (dt / args.steps) / scale / peak,
                "traffic": None,      # HBM counters need separate rocprofv3 --pmc passes: see profiled_* below
                "ms_per_launch": phases[dom]["ms_per_launch"],
                "concurrent_launch_sequences": parts if dom in hot else 1,
                "note": "achieved = algorithmic amount of the phase / its elapsed time (HIP events on the ctx stream, "
                        "all timed regions); ms_per_launch = that time / the phase's kernel launches"
                        + ("; the parts of the batch run their launch sequences concurrently, so one kernel's own "
                           "duration in a rocprofv3 trace is about concurrent_launch_sequences x ms_per_launch"
                           if (dom in hot and parts > 1) else "")}
        if default_cfg:
            # labelled as what they are: numbers of the committed rocprofv3 runs of this command, not of this run
            roof.update(committed_profile(ns_kernel.split("<")[0]))
            if "profiled_traffic_bytes" in roof:
                # HBM bytes per launch of the dominant kernel from the separate --pmc passes of the same command (FETCH_SIZE with
                # the guide's gfx950 correction for 16-byte-per-lane loads + WRITE_SIZE, tools/summarize_pmc.py): counters cannot
                # be collected inside this run, so the committed passes' figure stands here with its source beside it
                roof["traffic"] = roof["profiled_traffic_bytes"]
                roof["traffic_source"] = roof.get("profiled_traffic_source")
                roof["traffic_algorithmic_bytes"] = 3 * 8.0 * (Kl / max(parts, 1)) * p * p
        out = {
            "metric": "ADMM iters/sec on (K=32,p=500) GGL at 1/2/4/8 GPUs; eigh HBM GB/s vs peak", "value": its, "unit": "ADMM iters/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"SGL p={p}, {K}-point lambda1 grid logspace(0,-2) as one batch of {K} independent problems, "
                                    "identity start, rho0=1 update_rho per point, fixed iteration count" if is_grid else
                                    f"{reg} K={K} p={p} lambda1={l1} lambda2={l2} latent={latent}, identity start, "
                                    f"rho0=1 update_rho, fixed iteration count"),
                       "sharding": (f"K-slabs of {Kl} per GPU, collectives: " + ("RCCL behind the C ABI" if args.comm == "capi"
                                                                                  else "torch.distributed (RCCL)"))
                       if distributed else "single GPU",
                       "omega_step": "lds_jacobi" if eig_jacobi else ("newton_schulz_fp64_mfma" if omega_ns
                                                                         else "rocsolver_dsyevd+mfma_recon"),
                       # every ctx option as the library ran it (include/ggl_hip.h GGL_OPT_*): ns_tol is the relative
                       # spectral accuracy the Omega-step's matrix square root is iterated to (0 = fp64 resolution)
                       "options": effective_options, "options_overridden": sorted(options) or None,
                       **({"collectives_note": comm_note} if comm_note else {})},
            "timed_regions": {"count": len(region_s), "steps_each": args.steps, "statistic": "median",
                              "start_point": "identity start restored device-to-device before every region (no idle GPU "
                                             "between regions), then `warmup` untimed iterations",
                              "ms_per_step_min": min(region_s) / args.steps * 1e3,
                              "ms_per_step_max": max(region_s) / args.steps * 1e3},
            "roofline": roof,
            "iteration_hbm_roofline": {"algorithmic_bytes": iter_bytes, "achieved_GBs": its * iter_bytes / 1e9,
                                       "frac": its * iter_bytes / 1e9 / HBM_PEAK_GBS},
            "phases_ms": {ph: round(v["ms_per_launch"], 4) for ph, v in phases.items()},
            "phases_ms_per_step": {ph: round(per_iter(ph), 4) for ph in phases},
            "newton_schulz": {"steps_per_omega_step": (ns1["steps"] - ns0["steps"]) / max(1, ns1["calls"] - ns0["calls"]),
                              "stable_schedule_calls": ns1["stable_calls"] - ns0["stable_calls"],
                              "lstep_calls": ns1["rank_calls"] - ns0["rank_calls"],
                              "lstep_retries": ns1["rank_retries"] - ns0["rank_retries"],
                              "lstep_eigh_fallbacks": ns1["rank_fallbacks"] - ns0["rank_fallbacks"],
                              # two-tier L-step, totals since ctx creation (all regions, warm-up included): calls, calls whose
                              # coarse first pass was continued on a compact sub-batch, instances continued
                              "lstep_two_tier_totals": rank_totals,
                              "lstep_products_per_call": (ns1["rank_launches"] - ns0["rank_launches"])
                              / max(1, ns1["rank_calls"] - ns0["rank_calls"]) if latent else None,
                              "speculative_omega_steps": ns1["spec_calls"] - ns0["spec_calls"],
                              "speculation_misses": ns1["spec_misses"] - ns0["spec_misses"],
                              "prelaunched_chains_dropped": ns1["pre_dropped"] - ns0["pre_dropped"],
                              # totals since ctx creation: whole chains launched ahead / forgotten, early first parts / continued
                              "pipeline_totals": pipe_totals,
                              "end_of_iteration_poll_timeouts": ns1["spin_timeouts"] - ns0["spin_timeouts"],
                              # GGL_OPT_GROUP_SCHED (totals since ctx creation): Omega-steps that ran as contiguous groups with
                              # their own schedules, the split and the product units (A', B' included) of the last such step,
                              # mean units per group slot over the grouped steps
                              "group_schedules": (lambda g: {"grouped_steps": g["steps"], "last_groups": g["len"],
                                                             "last_products_per_group": g["units"],
                                                             "mean_products_per_group": [round(u / max(1, g["steps"]), 3)
                                                                                         for u in g["units_sum"][:max(len(g["len"]), 1)]]})
                              (group_totals) if group_totals else None}
            if omega_ns else None,
        }
        if exact is not None:
            out["value_exact"] = exact
        if distributed:
            out["n_ranks_seen"] = n_ranks_seen      # ncclCommCount of the engine's communicator (capi) / the process group's size
        if is_grid:
            out["grid_point_iterations_per_s"] = its * K
        if parity is not None:
            out["parity"] = parity
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
