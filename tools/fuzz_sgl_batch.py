#!/usr/bin/env python3
"""Random batches of single problems at p <= 64 (the one-launch iteration, k_omega_lds<.., SGL>) against the oracle's ADMM_SGL:
random p, K, lambda1 per point, shared / per-instance / no mask, padded instances of different dimension, a badly scaled
point now and then (the per-instance fallback).  Prints the worst deviation; exit code 1 on a mismatch.
    python tools/fuzz_sgl_batch.py [cases] [seed]"""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import synth
from gglasso_amd.batch import ADMM_SGL_batch, pad_blocks
from oracle import ggl_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad = 0.0, 0
for case in range(n_cases):
    p = int(rng.integers(2, 65))
    K = int(rng.integers(1, 13))
    S, _ = synth.make_problem("GGL", K, p, N=2 * p + 5, seed=int(rng.integers(1 << 30)))
    lam = np.exp(rng.uniform(np.log(0.02), np.log(0.6), K))
    kind = ["plain", "mask", "maskK", "dims", "scaled"][int(rng.integers(5))]
    kw = dict(tol=1e-8, rtol=1e-7, max_iter=400)
    mask = None
    ref_S, ref_lam, ref_mask, dims = [S[k] for k in range(K)], lam, [None] * K, None
    if kind == "mask":
        M = rng.uniform(0.2, 2.0, (p, p)); M = 0.5 * (M + M.T)
        lam = np.full(K, lam[0]); ref_lam = lam
        mask = M; ref_mask = [M] * K
    elif kind == "maskK":
        M = rng.uniform(0.2, 2.0, (K, p, p)); M = 0.5 * (M + M.transpose(0, 2, 1))
        mask = M; ref_mask = [M[k] for k in range(K)]
    elif kind == "dims" and p >= 4:
        dims = rng.integers(2, p + 1, K)
        ref_S = [S[k][:dims[k], :dims[k]] for k in range(K)]
        S = pad_blocks(ref_S, p, True)
    elif kind == "scaled":
        S = S.copy(); S[int(rng.integers(K))] *= float(rng.uniform(40, 120))
        ref_S = [S[k] for k in range(K)]
    args = dict(kw)
    if mask is not None:
        args["lambda1_mask"] = mask
    if dims is not None:
        args.update(dims=dims, Omega_0=pad_blocks([np.eye(d) for d in dims], p, True), X_0=np.zeros((K, p, p)))
    res = ADMM_SGL_batch(S, lam, **args)
    for k in range(K):
        q = ref_S[k].shape[0]
        with contextlib.redirect_stdout(io.StringIO()):
            ref, rinfo = orc.ADMM_SGL(ref_S[k], float(ref_lam[k]), np.eye(q), lambda1_mask=None if ref_mask[k] is None else ref_mask[k][:q, :q], **kw)
        d = float(np.abs(res[k][0]["Theta"] - ref["Theta"]).max()) / max(1.0, float(np.abs(ref["Theta"]).max()))
        worst = max(worst, d)
        ok = d <= 1e-7 and res[k][1]["status"] == rinfo["status"]
        if not ok:
            bad += 1
            print(f"case {case} ({kind}, p={p}, K={K}) point {k}: rel dev {d:.2e} status {res[k][1]['status']} / {rinfo['status']} iterations {res[k][1]['iterations']}")
print(f"{n_cases} cases, worst relative deviation of Theta {worst:.2e}, mismatches {bad}")
sys.exit(1 if bad else 0)
