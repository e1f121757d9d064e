"""TEST-ONLY stand-in for gglasso_amd.solver.HipEngine built on the CPU oracle, so that the host
logic (control flow of the ADMM loop, K-sharding, collectives) can be exercised without a GPU.
It lives under tests/ and is never importable from the product package."""
import numpy as np

from oracle import ggl_oracle as orc


class OracleEngine:
    def __init__(self, S, Omega_0, Theta_0, X_0, L_0=None, **_ignored):
        flat = lambda A: np.array(A, dtype=np.float64).reshape(-1, np.shape(A)[-2], np.shape(A)[-1])   # (G,K',p,p) -> (G*K',p,p)
        self.S = flat(S)
        self.K, self.p, _ = self.S.shape
        self.Om = flat(Omega_0)
        self.Om_prev = np.zeros_like(self.Om)
        self.Th = flat(Theta_0)
        self.X = flat(X_0)
        self.L = np.zeros_like(self.S) if L_0 is None else flat(L_0)
        self.groupsq = np.zeros((self.p, self.p))
        self.mask = None
        self.D = None
        self.beta = None

    def set_lambda1_mask(self, lam_pp):
        self.mask = lam_pp

    def set_lambda1_mask_k(self, lam_Kpp):
        self.maskK = lam_Kpp

    def set_instance_dims(self, pk):
        self.dims = None if pk is None else np.asarray(pk, dtype=int)

    def step_omega(self, rho, latent, nk):
        self._latent_seen = bool(latent)
        nk = np.ones(self.K) if nk is None else np.asarray(nk, dtype=np.float64)
        W = self.Th - self.L - self.X - (nk[:, None, None] / rho) * self.S
        self.Om_prev = self.Om
        self.beta = nk / rho
        self.Om, self.D = orc.phiplus_stack(W, self.beta)

    def step_group_partial(self, rho, lambda1):
        U = orc.prox_1norm(self.Om + self.L + self.X, (1 / rho) * lambda1)
        self.groupsq[...] = np.triu((U * U).sum(axis=0), 1)

    def step_finish(self, rho, lambda1, lambda2, reg, latent, mu1, groupsq_ready):
        V = self.Om + self.L + self.X
        l1, l2 = (1 / rho) * lambda1, (1 / rho) * lambda2
        if reg == 'SGL':
            lam = l1 if self.mask is None else (1 / rho) * self.mask
            self.Th = np.stack([orc.prox_od_1norm(V[k], lam) for k in range(self.K)])
        elif reg == 'GGL' and groupsq_ready:
            U = orc.prox_1norm(V, l1)
            a = np.maximum(np.sqrt(self.groupsq + self.groupsq.T), l2)
            Th = U * ((a - l2) / a)
            Th = np.triu(Th, 1)
            Th = Th + Th.transpose(0, 2, 1)
            d = np.arange(self.p)
            Th[:, d, d] = V[:, d, d]
            self.Th = Th
        else:
            self.Th = orc.prox_p(V, l1, l2, reg)
        if latent:
            self.L = orc.rank_stack(self.Th - self.X - self.Om, np.asarray(mu1) / rho)
        self.X = self.X + self.Om - self.Th + self.L
        return np.array([np.sum(self.Om ** 2), np.sum((self.Th - self.L) ** 2), np.sum(self.X ** 2),
                         np.sum((self.Om - self.Th + self.L) ** 2), np.sum((self.Om - self.Om_prev) ** 2)])

    def step(self, rho, lambda1, lambda2, reg, latent, mu1, nk):
        self.step_omega(rho, latent, nk)
        return self.step_finish(rho, lambda1, lambda2, reg, latent, mu1, 0)

    def scale_X(self, f):
        self.X = f * self.X

    # K independent single problems (batched lambda path)
    def sgl_batch_step(self, rho, lambda1, latent, mu1):
        self._latent_seen = bool(latent)
        rho = np.asarray(rho, dtype=np.float64)
        out = np.zeros((self.K, 5))
        Om_new = np.empty_like(self.Om)
        for k in range(self.K):
            W = self.Th[k] - self.L[k] - self.X[k] - (1 / rho[k]) * self.S[k]
            D, Q = np.linalg.eigh(W)
            om = orc.phiplus(1 / rho[k], D, Q)
            maskK = getattr(self, "maskK", None)
            lam = (1 / rho[k]) * lambda1[k] if self.mask is None else (1 / rho[k]) * self.mask
            if maskK is not None:
                lam = (1 / rho[k]) * maskK[k]
            th = orc.prox_od_1norm(om + self.L[k] + self.X[k], lam)
            if latent:
                C = th - self.X[k] - om
                D1, Q1 = np.linalg.eigh(C)
                self.L[k] = orc.prox_rank_norm(C, mu1[k] / rho[k], D1, Q1)
            x = self.X[k] + om - th + self.L[k]
            dims = getattr(self, "dims", None)
            q = self.p if dims is None else dims[k]        # padded instance: sums over its own block only
            b = lambda A: A[:q, :q]
            out[k] = [np.sum(b(om) ** 2), np.sum(b(th - self.L[k]) ** 2), np.sum(b(x) ** 2),
                      np.sum(b(om - th + self.L[k]) ** 2), np.sum(b(om - self.Om[k]) ** 2)]
            Om_new[k], self.Th[k], self.X[k] = om, th, x
        self.Om_prev, self.Om = self.Om, Om_new
        return out

    def scale_X_batch(self, factors):
        self.X = np.asarray(factors)[:, None, None] * self.X

    # G independent multiple-graph problems, problem g = instances g*K/G .. (batched lambda1 x lambda2 grid)
    def mgl_batch_step(self, G, rho, lambda1, lambda2, reg, latent, mu1, nk):
        self._latent_seen = bool(latent)
        Kp = self.K // G
        out = np.zeros((G, 5))
        Om_new = np.empty_like(self.Om)
        nkv = np.ones(Kp) if nk is None else np.asarray(nk, dtype=np.float64)
        for g in range(G):
            sl = slice(g * Kp, (g + 1) * Kp)
            W = self.Th[sl] - self.L[sl] - self.X[sl] - (nkv[:, None, None] / rho[g]) * self.S[sl]
            om, _ = orc.phiplus_stack(W, nkv / rho[g])
            th = orc.prox_p(om + self.L[sl] + self.X[sl], lambda1[g] / rho[g], lambda2[g] / rho[g], reg)
            if latent:
                self.L[sl] = orc.rank_stack(th - self.X[sl] - om, np.asarray(mu1)[sl] / rho[g])
            x = self.X[sl] + om - th + self.L[sl]
            out[g] = [np.sum(om ** 2), np.sum((th - self.L[sl]) ** 2), np.sum(x ** 2),
                      np.sum((om - th + self.L[sl]) ** 2), np.sum((om - self.Om[sl]) ** 2)]
            Om_new[sl], self.Th[sl], self.X[sl] = om, th, x
        self.Om_prev, self.Om = self.Om, Om_new
        return out

    def state_k(self, k, latent=False):
        sol = {'Omega': self.Om[k].copy(), 'Theta': self.Th[k].copy(), 'X': self.X[k].copy()}
        if latent:
            sol['L'] = self.L[k].copy()
        return sol

    def snapshot_k(self, k):
        if not hasattr(self, "_snapT"):
            self._snapT = np.zeros_like(self.Th)
        self._snapT[k] = self.Th[k]
        if getattr(self, "_latent_seen", False):
            if not hasattr(self, "_snapL"):
                self._snapL = np.zeros_like(self.L)
            self._snapL[k] = self.L[k]

    def finalize_L(self, which=0):
        # every L of this engine is an eigendecomposition's (orc.rank_stack): nothing to rebuild
        return 0, np.full(self.Th.shape[0], -1, dtype=np.int32)

    def snapshot_L_k(self, k):
        return self._snapL[k].copy()

    def threshold_scan(self, tau_range):
        tau_range = np.asarray(tau_range, dtype=np.float64)
        K = self.Th.shape[0]
        out = np.zeros((K, tau_range.size, 4))
        distinct = 0
        for k in range(K):
            counts = set()
            for j, tau in enumerate(tau_range):
                m = np.abs(self._snapT[k]) > tau
                np.fill_diagonal(m, True)
                T = self._snapT[k] * m
                d = np.linalg.eigvalsh(T)
                out[k, j] = [np.sum(self.S[k] * T), -np.inf if d.min() <= 1e-12 else np.linalg.slogdet(T)[1],
                             np.count_nonzero(T), d.min()]
                counts.add(int(out[k, j, 2]))
            distinct += len(counts)
        return out, distinct

    def selection_rank(self, rel_tol=0.0):
        K = self.Th.shape[0]
        rel = rel_tol if rel_tol > 0 else self.p * np.finfo(float).eps
        out = np.zeros((K, 4))
        for k in range(K):
            a = np.abs(np.linalg.eigvalsh(self._snapL[k]))
            keep = a > a.max() * rel
            out[k] = [keep.sum(), a.max(), a[~keep].max() if (~keep).any() else 0.0, a[keep].min() if keep.any() else 0.0]
        return out

    def selection_stats(self):
        K = self.Th.shape[0]
        out = np.zeros((K, 4))
        for k in range(K):
            T = self._snapT[k]
            d = np.linalg.eigvalsh(T)
            out[k] = [np.sum(self.S[k] * T), -np.inf if d.min() <= 1e-12 else np.linalg.slogdet(T)[1],
                      np.count_nonzero(T), d.min()]
        return out

    def objective(self, lambda1, lambda2, reg):
        ld = -np.log(orc.phip(self.D, self.beta[:, None])).sum()
        return np.array([ld, np.sum(self.Om * self.S), orc.P_val(self.Th, lambda1, lambda2, reg)])

    def kkt_residual(self, rho, lambda1, lambda2, reg, latent, mu1, nk):
        if reg == 'SGL':
            lam = lambda1 if self.mask is None else self.mask
            return orc.kkt_stopping_criterion_sgl(self.Om[0], self.Th[0], self.L[0], rho * self.X[0], self.S[0], lam,
                                                  latent, None if mu1 is None else mu1[0])
        return orc.kkt_stopping_criterion_mgl(self.Om, self.Th, self.L, rho * self.X, self.S, lambda1, lambda2,
                                              np.asarray(nk).reshape(-1, 1, 1), reg, latent, mu1)

    def exit_checks(self, latent):
        t = lambda A: np.abs(A - A.transpose(0, 2, 1)).max()
        return np.array([t(self.Om), t(self.Th), t(self.L), np.linalg.eigvalsh(self.Th - self.L).min(),
                         np.linalg.eigvalsh(self.L).min() if latent else 0.0])

    def exit_checks_k(self, latent):
        t = lambda A: np.abs(A - A.T).max()
        return np.array([[t(self.Om[k]), t(self.Th[k]), t(self.L[k]), np.linalg.eigvalsh(self.Th[k] - self.L[k]).min(),
                          np.linalg.eigvalsh(self.L[k]).min() if latent else 0.0] for k in range(self.K)])

    # ext_ADMM_MGL on the PADDED stacks (the layout of include/ggl_hip.h: identity block behind every instance), with
    # the oracle's per-instance operators: exercises the product's padding / un-padding and its host loop on the CPU
    def ext_setup(self, pk, G):
        self.pk = np.asarray(pk, dtype=int)
        self.G = np.asarray(G, dtype=int)
        self.Lam = None
        self.X1 = np.zeros_like(self.S)

    def ext_setup_batch(self, nprob, pk, G):
        self.nprob = int(nprob)
        self.pk = np.tile(np.asarray(pk, dtype=int), self.nprob)
        self.G = np.asarray(G, dtype=int)
        self.Lam = None
        self.X1 = np.zeros_like(self.S)

    def ext_batch_step(self, nprob, rho, lambda1K, lambda2G, latent, mu1):
        # problem by problem with the single-problem step on views of the slots (the products must agree with this)
        Kp = self.K // int(nprob)
        out = np.zeros((int(nprob), 5))
        full = (self.S, self.Om, self.Om_prev, self.Th, self.X, self.L, self.Lam, self.X1, self.pk, self.K)
        new = {nm: np.empty_like(self.S) for nm in ("Om", "Om_prev", "Th", "X", "L", "Lam", "X1")}
        for g in range(int(nprob)):
            sl = slice(g * Kp, (g + 1) * Kp)
            self.S, self.Om, self.Om_prev, self.Th, self.X, self.L, self.Lam, self.X1 = (A[sl].copy() for A in full[:8])
            self.pk, self.K = full[8][sl], Kp
            out[g] = self.ext_step(rho, np.asarray(lambda1K)[sl], float(lambda2G[g]), latent,
                                   None if mu1 is None else np.asarray(mu1)[sl])
            for nm in new:
                new[nm][sl] = getattr(self, nm)
        self.S, self.pk, self.K = full[0], full[8], full[9]
        for nm in new:
            setattr(self, nm, new[nm])
        return out

    def ext_set_state(self, Lambda, X1):
        self.Lam = np.array(Lambda, dtype=np.float64)
        self.X1 = np.zeros_like(self.S) if X1 is None else np.array(X1, dtype=np.float64)

    def ext_state(self):
        return {'Lambda': self.Lam.copy(), 'X1': self.X1.copy()}

    def _inner(self, A, k):
        return A[k, :self.pk[k], :self.pk[k]]

    def ext_step(self, rho, lambda1K, lambda2, latent, mu1):
        K = self.K
        self.step_omega(rho, latent, None)                   # block-diagonal: acts on data and padding separately
        V = (self.Om + self.L + self.X + self.Lam - self.X1) * 0.5
        self.Th = np.stack([orc.prox_od_1norm(V[k], lambda1K[k] / (2 * rho)) for k in range(K)])
        if latent:
            self.L = orc.rank_stack(self.Th - self.X - self.Om, np.asarray(mu1) / rho)
        Lam_prev = self.Lam
        Z = self.Th + self.X1
        self.Lam = np.stack([Z[k] for k in range(K)])
        shrunk = orc.prox_2norm_G({k: self._inner(Z, k) for k in range(K)}, self.G, lambda2 / rho)
        for k in range(K):
            self.Lam[k, :self.pk[k], :self.pk[k]] = shrunk[k]
        self.X = self.X + self.Om - self.Th + self.L
        self.X1 = self.X1 + self.Th - self.Lam
        n2 = lambda A: sum(np.sum(self._inner(A, k) ** 2) for k in range(K))
        return np.array([n2(self.Om) + n2(self.Lam), n2(self.Th - self.L) + n2(self.Th), n2(self.X) + n2(self.X1),
                         n2(self.Om - self.Th + self.L) + n2(self.Lam - self.Th),
                         n2(self.Om - self.Om_prev) + n2(self.Lam - Lam_prev)])

    def ext_kkt(self, rho, lambda1K, lambda2, latent, mu1):
        d = lambda A: {k: self._inner(A, k) for k in range(self.K)}
        return orc.ext_kkt_stopping_criterion(d(self.Om), d(self.Th), d(self.L), d(self.Lam), d(rho * self.X),
                                              d(rho * self.X1), d(self.S), self.G, lambda1K, lambda2, latent, mu1)

    def state(self):
        return {'Omega': self.Om.copy(), 'Theta': self.Th.copy(), 'L': self.L.copy(), 'X': self.X.copy()}

    def groupsq_tensor(self, torch, device):
        return torch.from_numpy(self.groupsq)      # shares memory with self.groupsq

    def groupsq_written(self, t):
        pass

    def sync(self):
        pass

    def close(self):
        pass


class SpeculatingOracleEngine(OracleEngine):
    """TEST-ONLY emulation of the HIP engine's K-sharded speculation protocol on the host, for world_size-2 gloo runs
    of the real driver loop (gglasso_amd.solver._run_admm): a speculative Omega-step may "miss" (scripted through
    GGL_TEST_MISS = "rank:call,rank:call"), which corrupts this rank's Omega and raises its flag; the flag rides
    on the group-sum all-reduce as element p*p; when the reduced flag is set every rank leaves its iterate alone,
    read_norms returns None and the driver repeats the step without speculation -- on all ranks, in lockstep."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        import os
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else 0
        self._miss_calls = {int(c) for r, c in (t.split(":") for t in os.environ.get("GGL_TEST_MISS", "").split(",") if t)
                            if int(r) == rank}
        self._spec_calls = 0
        self.retries = 0
        self._gsflat = np.zeros(self.p * self.p + 1)
        self._norms_dev = np.zeros(5)
        self._retry = False

    def step_omega(self, rho, latent, nk, speculate=False):
        self._saved = (self.Om, self.Om_prev)
        super().step_omega(rho, latent, nk)
        self._flag = 0.0
        if speculate:
            self._spec_calls += 1
            if self._spec_calls in self._miss_calls:
                self.Om = 1.5 * self.Om          # what a schedule built for too small a bound would deliver: garbage
                self._flag = 1.0

    def step_group_partial(self, rho, lambda1):
        super().step_group_partial(rho, lambda1)
        self._gsflat[:-1] = self.groupsq.ravel()
        self._gsflat[-1] = self._flag

    def groupsq_tensor(self, torch, device):
        return torch.from_numpy(self._gsflat)

    def norms_tensor(self, torch, device):
        return torch.from_numpy(self._norms_dev)

    def step_finish(self, rho, lambda1, lambda2, reg, latent, mu1, groupsq_ready, defer_norms=False):
        self.groupsq[...] = self._gsflat[:-1].reshape(self.p, self.p)
        if self._gsflat[-1] > 0.5:               # some rank missed: nobody touches Theta / X
            self._retry = True
            self._norms_dev[:] = np.nan
            return None
        sq = super().step_finish(rho, lambda1, lambda2, reg, latent, mu1, groupsq_ready)
        self._norms_dev[:] = sq
        return None if defer_norms else sq

    def read_norms(self):
        if self._retry:
            self._retry = False
            self.retries += 1
            self.Om, self.Om_prev = self._saved   # un-flip
            return None
        return self._norms_dev.copy()
