#!/usr/bin/env python3
"""Benchmark of the MI355X ADMM hot path: ADMM iterations/second on the Group Graphical Lasso
workload BASELINE.json quotes the metric on, (K=32, p=500), fp64.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full ADMM iteration (Omega-step: W formation + batched eigen/phiplus; Theta-step:
group prox over the (K,p,p) stack; dual update; the five stopping-test norms; rho rule on the host)
driven by the same host loop ``ADMM_MGL`` uses (gglasso_amd/solver.py::_run_admm) with tol=rtol=1e-20
so that it cannot exit early.  S, Omega, Theta, X are resident in HBM before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): the K=32 stack is sharded into K/N slabs
(strong scaling); per iteration one (p,p) fp64 all-reduce of the group sums of squares and one
5-scalar all-reduce of the residual norms go over RCCL.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the phase that dominates the iteration, timed live with HIP events on the ctx stream
  cpu_baseline -- the CPU oracle (NumPy/LAPACK + C prox) on this box's host cores, bounded sample
"""
import argparse
import contextlib
import io
import json
import os
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
}


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


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the PMC passes of this same command
    (tools/profile_round.sh -> tools/summarize_pmc.py -> profiles/*_pmc_summary.json; FETCH_SIZE and
    WRITE_SIZE are collected in separate rocprofv3 runs).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None
    data = json.load(open(files[-1]))
    key = kernel_name.split(" ")[0]
    for name, c in data.items():
        if name.startswith(key) and "hbm_bytes_per_launch" in c:
            return c["hbm_bytes_per_launch"]
    return None


def rocprof_kernel_avg_ms(kernel_name):
    """Average duration of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary of this
    same command (profiles/*_bench_kernel_stats.csv), for comparison with the live figure.  None when absent."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_kernel_stats.csv")))
    if not files:
        return None
    key = kernel_name.split(" ")[0]
    for row in csv.DictReader(open(files[-1])):
        if key in row["Name"]:
            return float(row["AverageNs"]) * 1e-6
    return None


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def cpu_baseline(S, reg, lambda1, lambda2, latent, mu1, iters):
    """CPU oracle (a port: NumPy eigh + C prox), bounded sample of the same workload."""
    from oracle import ggl_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:  # noqa: BLE001
        threads = os.cpu_count() or 1
    K, p, _ = S.shape
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    orc.ADMM_MGL(S, lambda1, lambda2, reg, Om0, max_iter=1, tol=1e-20, rtol=1e-20, latent=latent, mu1=mu1)
    t0 = time.perf_counter()
    orc.ADMM_MGL(S, lambda1, lambda2, reg, Om0, max_iter=iters, tol=1e-20, rtol=1e-20, latent=latent, mu1=mu1)
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "ADMM iters/s", "cores": int(threads), "kind": "port",
            "sample": f"{iters} ADMM iterations of the same ({reg}, K={K}, p={p}) problem from the identity start "
                      f"(oracle/ggl_oracle.py: numpy.linalg.eigh + C prox_p), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ggl_K32_p500", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eig", type=int, default=0, help="GGL_EIG_* selector (0 auto)")
    args = ap.parse_args()

    import torch
    from gglasso_amd import synth, solver, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GGL_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (1-GPU dev boxes)
    distributed = world > 1 or bool(os.environ.get("GGL_BENCH_FORCE_DIST"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    comm = None
    if distributed:
        if world == 1:      # GGL_BENCH_FORCE_DIST=1 without a launcher: single-rank rendezvous on the loopback
            for kk, vv in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                           ("MASTER_PORT", "29511")):
                os.environ.setdefault(kk, vv)
        import torch.distributed as dist
        from gglasso_amd.dist import TorchComm, shard_bounds
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:      # GGL_BENCH_FORCE_DIST=1 without a launcher: single-rank rendezvous on the loopback
            for kk, vv in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29511")):
                os.environ.setdefault(kk, vv)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        comm = TorchComm(device=f"cuda:{local_rank}")

    reg, K, p, latent, l1, l2, seed = WORKLOADS[args.workload]
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
    stream = comm.stream_handle if distributed else None      # the communicator's dedicated stream (gglasso_amd/dist.py)
    eng = solver.HipEngine(S_loc, Om0, Om0, np.zeros_like(S_loc), eig=args.eig, device=local_rank, stream=stream)
    nk = np.ones(Kl)
    mu_loc = None if mu1 is None else mu1[k0:k1]

    def run(iters, rho):
        info, rho = quiet(solver._run_admm, eng, reg, K, p, l1, l2, latent, mu_loc, nk, rho, 1e-20, 1e-20, 'boyd',
                          True, iters, False, False, "Multiple", comm=comm)
        return rho

    def fence():
        if distributed:
            torch.distributed.barrier()
        eng.sync()
        torch.cuda.synchronize()

    rho = 1.0
    if args.warmup > 0:
        rho = run(args.warmup, rho)
    ns0 = eng.ns_stats()
    # live HIP-event timing of the dominant (eigen / matrix-function) phases only during the timed region:
    # every extra event pair costs a few microseconds of host time per iteration
    eng.profile(2)
    eng.profile_read(reset=True)
    fence()
    t0 = time.perf_counter()
    rho = run(args.steps, rho)
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    prof = eng.profile_read(reset=True)
    ns1 = eng.ns_stats()
    # untimed extra pass with every phase instrumented, for the per-phase breakdown
    eng.profile(1)
    extra_iters = min(10, args.steps)
    run(extra_iters, rho)
    prof_all = eng.profile_read(reset=True)
    for ph, v in prof_all.items():
        if ph not in ("eig_omega", "eig_omega2", "eig_L"):
            prof[ph] = v
    eng.close()

    if rank == 0:
        eig_jacobi = (args.eig == _lib.EIG_JACOBI) or (args.eig == _lib.EIG_AUTO and p <= _lib.JACOBI_MAX_P)
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

        def per_iter(ph):   # hot phases were timed over the timed region, the others over the extra pass
            return phases[ph]["ms_per_launch"] * phases[ph]["launches"] / (args.steps if ph in hot else extra_iters)

        dom = max(phases, key=per_iter)
        omega_ns = (args.eig == _lib.EIG_NEWTON_SCHULZ) or (args.eig == _lib.EIG_AUTO and not eig_jacobi)
        bound, amount, unit = phase_model(dom, reg, Kl, p, latent, eig_jacobi, omega_ns)
        t64 = (p + 63) // 64
        ns_kernel = "k_symm_dl" if (p % 2 == 0 and (t64 * (t64 + 1) // 2 * Kl > 400 or p >= 384)) else "k_symm_tn"
        kernel_name = {"eig_omega": ns_kernel + " (Newton-Schulz product)" if omega_ns else
                       ("k_jacobi" if eig_jacobi else "rocsolver_dsyevd (library, many kernels)"),
                       "theta": "k_theta_ggl" if reg == "GGL" else ("k_theta_fgl" if reg == "FGL" else "k_theta_sgl"),
                       "recon_omega": "k_recon", "recon_L": "k_recon", "form_W": "k_form_W",
                       "dual": "k_dual_update", "eig_L": "k_jacobi" if eig_jacobi else "rocsolver_dsyevd"}.get(dom, dom)
        sec = phases[dom]["ms_per_launch"] * 1e-3
        if dom == "eig_L" and omega_ns and ns1["rank_launches"] > ns0["rank_launches"]:
            # L-step by sign Newton-Schulz: the phase is (2n+1) symmetric products of K p^3 flop each
            launches = ns1["rank_launches"] - ns0["rank_launches"]
            kernel_name = ns_kernel + " (sign Newton-Schulz product, L-step)"
            bound, unit = "mfma", "TFLOP/s"
            amount = 1.0 * Kl * p ** 3
            sec = phases[dom]["ms_per_launch"] * phases[dom]["launches"] * 1e-3 / launches
            phases[dom] = {"ms_per_launch": sec * 1e3, "launches": launches}
        if dom == "eig_omega" and omega_ns:
            # the Omega-step's launches differ in size (a pair launch carries two products): average over
            # the step = algorithmic flop of all its launches / their total duration (HIP events)
            units = ns1["units"] - ns0["units"]
            launches = ns1["launches"] - ns0["launches"]
            amount = units * 1.0 * Kl * p ** 3 / launches
        if bound == "hbm":
            achieved, peak = amount / sec / 1e9, HBM_PEAK_GBS
        else:
            achieved, peak = amount / sec / 1e12, FP64_MFMA_PEAK_TF
        its = args.steps / dt
        # the committed rocprofv3 / PMC summaries are of the default command only
        profiled_cfg = (args.workload == "ggl_K32_p500" and args.eig == 0 and not os.environ.get("GGL_NS_MODE"))
        iter_bytes = (120.0 if latent else 72.0) * Kl * p * p          # SURVEY.md 8(d), per GPU
        out = {
            "metric": "ADMM iters/sec on (K=32,p=500) GGL at 1/2/4/8 GPUs; eigh HBM GB/s vs peak", "value": its, "unit": "ADMM iters/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{reg} K={K} p={p} lambda1={l1} lambda2={l2} latent={latent}, identity start, "
                                   f"rho0=1 update_rho, fixed iteration count",
                       "sharding": f"K-slabs of {Kl} per GPU" if distributed else "single GPU",
                       "omega_step": "lds_jacobi" if eig_jacobi else ("newton_schulz_fp64_mfma" if omega_ns
                                                                         else "rocsolver_dsyevd+mfma_recon")},
            "roofline": {"kernel": kernel_name, "phase": dom, "launches_per_step": phases[dom]["launches"] / args.steps,
                         "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": pmc_traffic(kernel_name) if profiled_cfg else None,
                         "ms_per_launch": phases[dom]["ms_per_launch"],
                         "rocprof_kernel_avg_ms": rocprof_kernel_avg_ms(kernel_name) if profiled_cfg else None,
                         "concurrent_launch_sequences": 2 if (omega_ns and dom in ("eig_omega", "eig_L") and Kl >= 16
                                                              and 600 <= t64 * (t64 + 1) // 2 * Kl <= 2048) else 1,
                         "note": ("ms_per_launch = elapsed time of the phase / its kernel launches; at this size two parts "
                                  "of the batch run their launch sequences concurrently on two streams, so a single "
                                  "kernel's own duration (rocprof_kernel_avg_ms, rocprofv3) is about "
                                  "concurrent_launch_sequences x ms_per_launch") if (omega_ns and dom == "eig_omega"
                                                                                        and Kl >= 16 and 600 <= t64 * (t64 + 1) // 2 * Kl <= 2048)
                         else "elapsed time of the phase / its kernel launches"},
            "iteration_hbm_roofline": {"algorithmic_bytes": iter_bytes, "achieved_GBs": its * iter_bytes / 1e9,
                                       "frac": its * iter_bytes / 1e9 / HBM_PEAK_GBS},
            "phases_ms": {ph: round(v["ms_per_launch"], 4) for ph, v in phases.items()},
            "phases_ms_per_step": {ph: round(per_iter(ph), 4) for ph in phases},
            "newton_schulz": {"steps_per_omega_step": (ns1["steps"] - ns0["steps"]) / max(1, ns1["calls"] - ns0["calls"]),
                              "stable_schedule_calls": ns1["stable_calls"] - ns0["stable_calls"],
                              "lstep_calls": ns1["rank_calls"] - ns0["rank_calls"],
                              "lstep_retries": ns1["rank_retries"] - ns0["rank_retries"],
                              "lstep_eigh_fallbacks": ns1["rank_fallbacks"] - ns0["rank_fallbacks"],
                              "speculative_omega_steps": ns1["spec_calls"] - ns0["spec_calls"],
                              "speculation_misses": ns1["spec_misses"] - ns0["spec_misses"]} if omega_ns else None,
        }
        if not distributed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, reg, l1, l2, latent, mu1, args.cpu_iters)
        print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
