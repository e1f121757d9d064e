"""Multi-GPU execution: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI).

What shards and what does not:
  * Omega-step / L-step (eigendecompositions): independent per instance k -> contiguous K-slabs.
  * GGL Theta-step: the group norm |u[:,i,j]|_2 runs over ALL K (solver/ggl_helper.py:38-43), so a
    K-sharded run needs one exchange per iteration: every rank soft-thresholds its slab, accumulates
    sum_k u^2 into the packed upper triangle of a (p,p) buffer, the buffers are all-reduced (4*p*(p+1)+8 bytes; 1 MB at
    p=500), and every
    rank scales its own slab.  This is the only data-path collective.
  * stopping test: the five squared norms are all-reduced (40 bytes) before the host takes the sqrt.
  * FGL Theta-step: Condat's scan is sequential along K (solver/fgl_helper.py:24-66) -> no K-sharding;
    FGL scales through independent replicas (model-selection grid points), see ``shard_grid``.

The reference has no distributed code at all; there is nothing to translate.
"""
import numpy as np

from . import _lib
from .solver import _run_admm, _exit_report, as_c
from . import solver as _solver


def shard_bounds(K, world, rank):
    """Contiguous, balanced K-slab [k0, k1) of ``rank``."""
    base, rem = divmod(K, world)
    k0 = rank * base + min(rank, rem)
    return k0, k0 + base + (1 if rank < rem else 0)


def shard_grid(n_points, world, rank):
    """Indices of the model-selection grid points (independent solves) owned by ``rank``:
    round-robin, so every rank gets a mix of cheap (large lambda) and expensive (small lambda) points."""
    return list(range(rank, n_points, world))


class _DeviceView:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class TorchComm:
    """All-reduces over a torch.distributed process group.  With the nccl backend the tensors stay in
    HBM and the collective is RCCL; with gloo (CPU tests) they are host tensors.

    Stream discipline (nccl): the engine's kernels and the collectives must be ordered on ONE stream -- RCCL orders
    a collective against the stream that is current when it is issued, and nothing else orders it against kernels on
    another stream.  The communicator therefore owns a dedicated ``torch.cuda.Stream`` (``self.stream``), the ctx is
    created on that stream's (non-NULL) handle, and every collective is issued under ``torch.cuda.stream(self.stream)``
    after checking that the engine really runs on it."""

    def __init__(self, group=None, device=None, stream=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.backend = dist.get_backend(group)
        self.device = device
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._gs = None
        self._nrm = None
        self.device_norms = self.backend == "nccl"      # all-reduce the norms where the kernels leave them
        self.stream = None
        if self.backend == "nccl":
            self.stream = stream if stream is not None else torch.cuda.Stream(device=device)

    @property
    def stream_handle(self):
        """HIP stream handle to create the engine on (None for host backends)."""
        return None if self.stream is None else int(self.stream.cuda_stream)

    def _on_stream(self, eng=None):
        import contextlib
        if self.stream is None:
            return contextlib.nullcontext()
        if eng is not None and getattr(eng, "stream_handle", None) != self.stream_handle:
            raise RuntimeError("TorchComm: the engine does not run on the communicator's stream "
                               f"(engine {getattr(eng, 'stream_handle', None)}, comm {self.stream_handle}): the collective "
                               "would not be ordered against its kernels")
        return self.torch.cuda.stream(self.stream)

    def allreduce_groupsq(self, eng):
        # the tensor aliases a buffer of THIS engine's ctx: one comm may serve several solves
        if self._gs is None or self._gs[0] is not eng:
            self._gs = (eng, eng.groupsq_tensor(self.torch, self.device))
        with self._on_stream(eng):
            self.dist.all_reduce(self._gs[1], op=self.dist.ReduceOp.SUM, group=self.group)
        eng.groupsq_written(self._gs[1])

    def allreduce_norms_device(self, eng):
        if self._nrm is None or self._nrm[0] is not eng:
            self._nrm = (eng, eng.norms_tensor(self.torch, self.device))
        with self._on_stream(eng):
            self.dist.all_reduce(self._nrm[1], op=self.dist.ReduceOp.SUM, group=self.group)
        return eng.read_norms()

    def allreduce_norms(self, arr):
        t = self.torch.as_tensor(np.asarray(arr, dtype=np.float64))
        with self._on_stream():
            if self.backend == "nccl":
                t = t.to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()


class CommIdError(RuntimeError):
    """RCCL's unique id could not be created on rank 0; raised on every rank of the group by ``RcclComm.attach``."""


class RcclComm:
    """RCCL behind the C ABI (include/ggl_hip.h, ggl_comm_* / ggl_admm_step_sharded): the library itself issues the
    two all-reduces of a K-sharded GGL iteration on the ctx stream, so an iteration is ONE C call and the host
    program needs no torch tensors at all.  torch.distributed is used here for exactly one thing: shipping RCCL's
    128-byte unique id from rank 0 to the other ranks (any broadcast would do: MPI, a file, a socket)."""
    capi = True
    backend = "rccl-capi"
    device_norms = True

    def __init__(self, group=None, device=None):
        """device: the GPU index of this rank's engine.  torch's own collectives below (the object broadcast of the
        unique id, the host-side sum) run on torch's CURRENT device, which is cuda:0 in every rank of a caller that never
        called torch.cuda.set_device -- so they are issued under ``torch.cuda.device(device)``; None: taken from the
        engine in ``attach``."""
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = None if device is None else int(device)
        self.stream_handle = None           # the ctx creates its own stream; the collectives are issued on it in C

    def _on_device(self):
        import contextlib
        if self.dist.get_backend(self.group) != "nccl" or self.device is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.device(self.device)

    def attach(self, eng):
        """Create this rank's communicator inside the engine's ctx (collective: every rank calls it).
        A failure to obtain the unique id on rank 0 travels with the broadcast and is raised as ``CommIdError`` on EVERY
        rank (so a caller may fall back collectively); a failure inside ``ncclCommInitRank`` on some rank only cannot be
        made symmetric from here -- the other ranks are inside RCCL's own rendezvous -- and such a rank should leave the
        job (a non-zero exit makes the launcher tear the others down) rather than enter another collective."""
        import ctypes
        if self.device is None:
            self.device = getattr(eng, "device", None)
        box = [None]
        if self.rank == 0:
            try:
                buf = ctypes.create_string_buffer(128)
                _lib.check(_lib.load().ggl_comm_unique_id(buf))
                box[0] = buf.raw
            except Exception as e:  # noqa: BLE001 -- reported to every rank below
                box[0] = ("error", f"{type(e).__name__}: {e}")
        with self._on_device():
            self.dist.broadcast_object_list(box, src=0, group=self.group)
        if isinstance(box[0], tuple):
            raise CommIdError(f"rank 0 could not create the RCCL unique id ({box[0][1]})")
        eng.comm_init(self.rank, self.world, box[0])

    def allreduce_norms(self, arr):
        """Host-side sum over ranks (objective values with measure=True; not on the iteration path)."""
        import torch
        t = torch.as_tensor(np.asarray(arr, dtype=np.float64))
        with self._on_device():
            if self.dist.get_backend(self.group) == "nccl":
                t = t.to("cuda" if self.device is None else f"cuda:{self.device}")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()


def _hip_groupsq_tensor(self, torch, device):
    # the packed upper triangle of the (p,p) sums + the trailing speculation flag (include/ggl_hip.h, GGL_BUF_GROUPSQ):
    # one flat vector of p (p + 1) / 2 + 1 doubles
    ptr = self.device_ptr(_lib.BUF_GROUPSQ)
    return torch.as_tensor(_DeviceView(ptr, (self.p * (self.p + 1) // 2 + 1,)), device=device)


def _hip_norms_tensor(self, torch, device):
    return torch.as_tensor(_DeviceView(self.device_ptr(_lib.BUF_NORMS), (5,)), device=device)


def _hip_groupsq_written(self, t):
    pass    # the tensor aliases the ctx buffer; TorchComm issued the collective on the ctx's own stream => ordered


_solver.HipEngine.groupsq_tensor = _hip_groupsq_tensor
_solver.HipEngine.groupsq_written = _hip_groupsq_written
_solver.HipEngine.norms_tensor = _hip_norms_tensor


def ADMM_MGL_sharded(S_local, lambda1, lambda2, reg, Omega_0, K_total, comm, Theta_0=np.array([]),
                     X_0=np.array([]), n_samples=None, tol=1e-5, rtol=1e-4, update_rho=True, rho=1.,
                     max_iter=1000, verbose=False, measure=False, device=0, engine_kwargs=None, latent=False, mu1=None):
    """K-sharded Group Graphical Lasso: this rank owns the slab ``S_local`` (K_local,p,p) of a problem
    with ``K_total`` instances; arguments otherwise as ADMM_MGL (solver/admm_solver.py:13-31).  Every
    rank executes the same host loop and sees the same residuals, so the rho updates and the stopping
    decision agree without any further communication.  Returns this rank's slab of the solution.
    ``latent`` / ``mu1`` (this rank's (K_local,) slab of it, or a float): the L-step (admm_solver.py:197-205) is per
    instance and runs on the local slab; the group sums use Omega + L + X."""
    assert reg == 'GGL', "only the GGL penalty shards across K (FGL scans along K; use grid sharding)"
    assert Omega_0.shape == S_local.shape
    assert min(lambda1, lambda2) > 0
    assert rho > 0
    Kl, p, _ = S_local.shape
    if latent:
        if isinstance(mu1, float):
            mu1 = mu1 * np.ones(Kl)
        assert mu1 is not None
        assert np.all(mu1 > 0)
        mu1 = as_c(mu1)
        assert len(mu1) == Kl
    else:
        mu1 = None
    if n_samples is None:
        nk = np.ones(Kl)
    elif isinstance(n_samples, (int, np.integer)):
        nk = float(n_samples) * np.ones(Kl)
    else:
        nk = as_c(n_samples).reshape(-1)
        assert len(nk) == Kl
    if len(Theta_0) == 0:
        Theta_0 = Omega_0
    if len(X_0) == 0:
        X_0 = np.zeros((Kl, p, p))
    kw = dict(engine_kwargs or {})
    if getattr(comm, "capi", False):
        kw.setdefault("device", device)
    elif comm.backend == "nccl":
        # the ctx runs on the communicator's dedicated stream: RCCL orders a collective against the stream it is
        # issued on, so kernels and all-reduces need no host synchronisation on either side (TorchComm checks it)
        kw.setdefault("stream", comm.stream_handle)
        kw.setdefault("device", device)
    eng = _solver.ENGINE(S_local, Omega_0, Theta_0, X_0, **kw)
    try:
        if getattr(comm, "capi", False):
            comm.attach(eng)
        info, _ = _run_admm(eng, reg, K_total, p, float(lambda1), float(lambda2), bool(latent), mu1, nk, float(rho),
                            tol, rtol, 'boyd', update_rho, max_iter, verbose and comm.rank == 0, measure,
                            "Multiple", comm=comm, want_objective=False)
        if latent and hasattr(eng, "finalize_L"):
            eng.finalize_L()        # per instance, on this rank's slab: no exchange
        _exit_report(eng, bool(latent), 1e-5, False)
        sol = eng.state()
    finally:
        eng.close()
    return sol, info


def lambda_path_sharded(S, lambda1, group=None, **kwargs):
    """The lambda1 path of a Single Graphical Lasso problem (the outer loop of the reference's
    ``single_grid_search``, helper/model_selection.py:619-630) spread over the ranks of a process group:
    rank r solves the grid points ``shard_grid(len(lambda1), world, r)`` as ONE batch on its GPU
    (gglasso_amd.batch.ADMM_SGL_batch) and the (sol, info) pairs are gathered on every rank, in grid order.
    Grid points are independent problems, so there is no data-path collective -- replicas only.
    ``kwargs`` go to ``ADMM_SGL_batch`` (tol, rtol, latent, mu1, ...)."""
    import torch.distributed as dist
    from .batch import ADMM_SGL_batch
    lam = np.atleast_1d(np.asarray(lambda1, dtype=np.float64))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = shard_grid(len(lam), world, rank)
    local = ADMM_SGL_batch(S, lam[mine], **kwargs) if mine else []
    gathered = [None] * world
    dist.all_gather_object(gathered, list(zip(mine, local)), group=group)
    out = [None] * len(lam)
    for part in gathered:
        for idx, res in part:
            out[idx] = res
    return out
