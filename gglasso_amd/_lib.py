"""ctypes binding of libggl_hip.so (include/ggl_hip.h).  There is no fallback: if the library is
missing, fails to load, or sees no GPU when one is required, this module raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libggl_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "lib", "libggl_hip_dev.so")     # python -m gglasso_amd.build --dev (tools/ only)

# mirrors include/ggl_hip.h
REG_SGL, REG_GGL, REG_FGL = 0, 1, 2
EIG_AUTO, EIG_JACOBI, EIG_ROCSOLVER, EIG_NEWTON_SCHULZ = 0, 1, 2, 3
CTX_STREAM_GIVEN = 1 << 16
# ggl_ctx_set_option ids (GGL_OPT_*)
OPTIONS = {"speculate": 1, "spec_factor": 2, "ns_mode": 3, "ns_degrees": 4, "theta_flat": 5, "rank_eig": 6, "parts": 7,
           "parts_max_tiles": 8, "symm_variant": 9, "spin_wait": 10, "fused_bounds": 11, "pipeline": 12, "fused_start": 13, "parts_small": 14, "ns_tol": 15, "cw_warm": 16, "chain": 17, "rank_l0_coarse": 18, "isolate": 19, "rank_deflate": 20, "rank_l0_deflate": 21, "fused_cw": 22, "omega_lds": 23, "early_part": 24, "part_priority": 25, "fused_w": 26, "rank_cw": 27, "bound_side": 28, "lds_pinned": 29, "join_flag": 30, "cw_rider": 31, "copy_rider": 32, "reduce_rider": 33, "parts_bias": 34, "parts_order": 35, "download_threads": 36, "group_sched": 37}


def eig_flags(method=EIG_AUTO, ns_mode=0, ns_degrees=0):
    """eigensolver selector with the Newton-Schulz controls in the upper bits (GGL_EIG_NS_MODE / GGL_EIG_NS_DEGREES)."""
    return int(method) | ((int(ns_mode) & 0x3) << 8) | ((int(ns_degrees) & 0xf) << 12)

JACOBI_MAX_P = 128
NS_MIN_P = 8          # GGL_EIG_AUTO: matrix-function Omega- / L-step for p above this (include/ggl_hip.h GGL_NS_MIN_P)
BUF_S, BUF_OMEGA, BUF_THETA, BUF_L, BUF_X, BUF_GROUPSQ, BUF_NORMS, BUF_OMEGA_PREV = range(8)
E_ARG, E_HIP, E_SOLVER, E_ALLOC, E_COMM = -1, -2, -3, -4, -5
PHASES = ("form_W", "eig_omega", "recon_omega", "theta", "eig_L", "recon_L", "dual", "reduce", "eig_omega2", "bound",
          "allreduce_groupsq", "allreduce_norms")

_dp = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double
_ubp = ctypes.POINTER(ctypes.c_ubyte)
_ip = ctypes.POINTER(ctypes.c_int)

_SIGNATURES = {
    "ggl_version": ([], _i),
    "ggl_last_error": ([], ctypes.c_char_p),
    "ggl_device_count": ([], _i),
    "ggl_theta_limits": ([ctypes.POINTER(_i)], _i),
    "ggl_ctx_create": ([_i, _i, _i, _i, _vp, ctypes.POINTER(_vp)], _i),
    "ggl_ctx_destroy": ([_vp], _i),
    "ggl_ctx_sync": ([_vp], _i),
    "ggl_device_ptr": ([_vp, _i], _vp),
    "ggl_ctx_set_option": ([_vp, _i, _d], _i),
    "ggl_ctx_get_option": ([_vp, _i, _dp], _i),
    "ggl_set_S": ([_vp, _dp], _i),
    "ggl_set_state": ([_vp, _dp, _dp, _dp, _dp], _i),
    "ggl_set_state_ex": ([_vp, _dp, _dp, _dp, _dp, ctypes.POINTER(_i)], _i),
    "ggl_set_S_ex": ([_vp, _dp, _i], _i),
    "ggl_get_state": ([_vp, _dp, _dp, _dp, _dp], _i),
    "ggl_state_snapshot": ([_vp, _i], _i),
    "ggl_set_lambda1_mask": ([_vp, _dp], _i),
    "ggl_set_lambda1_mask_k": ([_vp, _dp], _i),
    "ggl_set_instance_dims": ([_vp, ctypes.POINTER(_i)], _i),
    "ggl_admm_step": ([_vp, _d, _d, _d, _i, _i, _dp, _dp, _dp], _i),
    "ggl_hint_last_step": ([_vp], _i),
    "ggl_step_omega": ([_vp, _d, _i, _dp], _i),
    "ggl_step_omega_spec": ([_vp, _d, _i, _dp], _i),
    "ggl_step_group_partial": ([_vp, _d, _d], _i),
    "ggl_step_finish": ([_vp, _d, _d, _d, _i, _i, _dp, _i, _dp], _i),
    "ggl_norms_read": ([_vp, _dp], _i),
    "ggl_comm_unique_id": ([ctypes.c_char_p], _i),
    "ggl_comm_init": ([_vp, _i, _i, ctypes.c_char_p], _i),
    "ggl_comm_destroy": ([_vp], _i),
    "ggl_comm_count": ([_vp, ctypes.POINTER(_i)], _i),
    "ggl_allreduce_groupsq": ([_vp], _i),
    "ggl_allreduce_norms": ([_vp], _i),
    "ggl_admm_step_sharded": ([_vp, _d, _d, _d, _dp, _dp], _i),
    "ggl_admm_step_sharded_latent": ([_vp, _d, _d, _d, _i, _dp, _dp, _dp], _i),
    "ggl_scale_X": ([_vp, _d], _i),
    "ggl_sgl_batch_step": ([_vp, _dp, _dp, _i, _dp, _dp], _i),
    "ggl_mgl_batch_step": ([_vp, _i, _dp, _dp, _dp, _i, _i, _dp, _dp, _dp], _i),
    "ggl_scale_X_batch": ([_vp, _dp], _i),
    "ggl_batch_decide": ([_i, _dp, _ubp, _ubp, _dp, _dp, _d, _d, _i, _dp, _dp, _ip], _i),
    "ggl_sgl_batch_run": ([_vp, _i, _dp, _dp, _i, _dp, _dp, _d, _d, _i, _dp, _ip, _ip, _i, _vp, _ip, _i], _i),
    "ggl_mgl_batch_run": ([_vp, _i, _i, _dp, _dp, _dp, _i, _i, _dp, _dp, _dp, _d, _d, _i, _dp, _ip, _ip, _i, _vp, _ip, _i], _i),
    "ggl_snapshot_state_from": ([_vp, _i, _vp, _i], _i),
    "ggl_get_snapshots": ([_vp, _dp, _dp, _dp, _dp], _i),
    "ggl_get_state_k": ([_vp, _i, _dp, _dp, _dp, _dp], _i),
    "ggl_exit_checks": ([_vp, _i, _dp], _i),
    "ggl_exit_checks_k": ([_vp, _i, _dp], _i),
    "ggl_exit_checks_fast_k": ([_vp, _i, _d, _d, _dp], _i),
    "ggl_ext_setup": ([_vp, ctypes.POINTER(_i), ctypes.POINTER(_i), _i], _i),
    "ggl_ext_set_state": ([_vp, _dp, _dp], _i),
    "ggl_ext_get_state": ([_vp, _dp, _dp], _i),
    "ggl_ext_admm_step": ([_vp, _d, _dp, _d, _i, _dp, _dp], _i),
    "ggl_ext_kkt_residual": ([_vp, _d, _dp, _d, _i, _dp, _dp], _i),
    "ggl_ext_setup_batch": ([_vp, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), _i], _i),
    "ggl_ext_batch_step": ([_vp, _d, _dp, _dp, _i, _dp, _dp], _i),
    "ggl_objective": ([_vp, _d, _d, _i, _dp], _i),
    "ggl_kkt_residual": ([_vp, _d, _d, _d, _i, _i, _dp, _dp, _dp], _i),
    "ggl_profile_enable": ([_vp, _i], _i),
    "ggl_profile_read": ([_vp, _dp, ctypes.POINTER(ctypes.c_longlong), _i], _i),
    "ggl_dev_symm": ([_i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _i], _i),
    "ggl_dev_symm_bench": ([_i, _i, _i, _i, _dp], _i),
    "ggl_lds_stats": ([_vp, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_group_stats": ([_vp, ctypes.POINTER(ctypes.c_longlong), _dp], _i),
    "ggl_spectral_bounds": ([_vp, _dp, _dp], _i),
    "ggl_pipeline_stats": ([_vp, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_trace_start": ([_vp, _i], _i),
    "ggl_trace_read": ([_vp, _dp, _i], _i),
    "ggl_dev_omega_lds": ([_i, _i, _dp, _dp, _dp, _dp, _dp, _d, _i, _dp, _dp, _i, _dp], _i),
    "ggl_dev_symm_bounds": ([_i, _i, _dp, _dp, _i, _dp, _dp, _dp, _dp], _i),
    "ggl_snapshot_k": ([_vp, _i], _i),
    "ggl_snapshot_from": ([_vp, _i, _vp, _i], _i),
    "ggl_selection_stats": ([_vp, _dp], _i),
    "ggl_threshold_scan": ([_vp, _dp, _i, _dp, ctypes.POINTER(_i)], _i),
    "ggl_selection_rank": ([_vp, _d, _dp], _i),
    "ggl_finalize_L": ([_vp, _i, ctypes.POINTER(_i)], _i),
    "ggl_failed_instances": ([_vp, ctypes.POINTER(_i)], _i),
    "ggl_failed_reason": ([_vp, _i, _dp], _i),
    "ggl_debug_poison": ([_i], _i),
    "ggl_set_odd_dl": ([_i], _i),
    "ggl_reset_instance": ([_vp, _i], _i),
    "ggl_ctx_create_subset": ([_vp, ctypes.POINTER(_i), _i, ctypes.POINTER(_vp)], _i),
    "ggl_get_snapshot_k": ([_vp, _i, _dp, _dp], _i),
    "ggl_get_snapshot_state_k": ([_vp, _i, _dp, _dp], _i),
    "ggl_dev_ns_schedule": ([_d, _i, _i, ctypes.POINTER(_i), _dp, ctypes.POINTER(_i)], _i),
    "ggl_dev_ns_schedule_tol": ([_d, _i, _d, _i, ctypes.POINTER(_i), _dp, ctypes.POINTER(_i)], _i),
    "ggl_dev_group_partition": ([ctypes.POINTER(_i), _i, _i, _i, ctypes.POINTER(_i)], _i),
    "ggl_dev_ns_units": ([_d, _i, _d], _i),
    "ggl_ns_stats": ([_vp, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_rank_stats": ([_vp, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_deflate_stats": ([_vp, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_finalize_stats": ([_vp, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_eig_info": ([_vp, ctypes.POINTER(_i)], _i),
    "ggl_last_dispatch": ([_vp, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_eigh_batched": ([_i, _i, _dp, _dp, _dp, _i], _i),
    "ggl_phiplus": ([_i, _i, _dp, _dp, _dp, _dp], _i),
    "ggl_prox_rank_norm": ([_i, _i, _dp, _dp, _dp, _dp], _i),
    "ggl_phiplus_matrix": ([_i, _i, _dp, _dp, _dp, _i], _i),
    "ggl_rank_matrix": ([_i, _i, _dp, _dp, _dp, _i], _i),
    "ggl_rank_matrix_ex": ([_i, _i, _dp, _dp, _dp, _i, _d, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_rank_matrix_deflate": ([_i, _i, _dp, _dp, _dp, _i, _d, ctypes.POINTER(ctypes.c_longlong)], _i),
    "ggl_prox_od_1norm": ([_i, _dp, _d, _dp, _dp], _i),
    "ggl_prox_p": ([_i, _i, _dp, _d, _d, _i, _dp], _i),
    "ggl_prox_tv": ([_i, _i, _dp, _d, _dp], _i),
    "ggl_prox_2norm": ([_i, _i, _dp, _d, _dp], _i),
    "ggl_prox_phi": ([_i, _i, _dp, _d, _d, _i, _dp], _i),
}

# libggl_hip_dev.so only (-DGGL_DEV)
_DEV_SIGNATURES = {
    "ggl_dev_i8_stages": ([_i], _i),
    "ggl_dev_symm_i8": ([_i, _i, _i, _i, _dp, _dp, _d, _d, _dp, _i, _dp], _i),
    "ggl_dev_omega_i8": ([_i, _i, _dp, _dp, _dp, ctypes.POINTER(_i), _d, _dp, _i, _dp], _i),
    "ggl_dev_mfma_f64_peak": ([_dp], _i),
    "ggl_dev_symm_timeline": ([_i, _i, ctypes.POINTER(ctypes.c_longlong), _i, ctypes.POINTER(_i)], _i),
    "ggl_dev_chain_probe": ([_i, _i, _i, _i, _i, _i, _dp], _i),
    "ggl_dev_coissue_probe": ([_dp], _i),
    "ggl_dev_mfma_lds_probe": ([_dp], _i),
    "ggl_dev_chain_run": ([_i, _i, _i, _i, _dp], _i),
    "ggl_dev_switch_bench": ([_i, _i, _i, _i, _i, _dp], _i),
    "ggl_dev_vendor_bench": ([_i, _i, _i, _i, _dp], _i),
    "ggl_dev_fill_copy_probe": ([_i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_longlong)], _i),
}

ABI_VERSION = 300      # include/ggl_hip.h GGL_VERSION: argument layouts and buffer formats this binding was written against
EXPORTS = tuple(_SIGNATURES)
_lib = None


def load():
    """Load libggl_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m gglasso_amd.build` "
            "(gglasso_amd has no CPU fallback)")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 / rocBLAS / rocSOLVER under the
    # same SONAMEs as /opt/rocm.  Whichever copy is loaded first serves the whole process; if the system
    # copies come first and torch is imported later, torch's second HSA runtime sees no GPU ("No HIP GPUs are
    # available").  The multi-GPU path (gglasso_amd.dist) needs torch in the same process, so torch -- when it
    # is installed -- is imported before the library, once (set GGL_TORCH_FIRST=0 to skip).
    import sys
    if "torch" not in sys.modules and os.environ.get("GGL_TORCH_FIRST", "1") != "0":
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001  (torch absent or broken: the single-GPU path does not need it)
            pass
    lib = _bind(ctypes.CDLL(LIB_PATH), _SIGNATURES)
    v = lib.ggl_version()
    if v // 100 != ABI_VERSION // 100:
        raise RuntimeError(f"{LIB_PATH} speaks ABI version {v}, this binding {ABI_VERSION} (include/ggl_hip.h GGL_VERSION): "
                           "rebuild it with `python -m gglasso_amd.build --force`")
    _lib = lib
    return _lib


def _bind(lib, table):
    for name, (argtypes, restype) in table.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


def load_dev():
    """The development build (ablation variants, probes, GGL_* environment knobs) for tools/; never used by the
    solvers.  Build it with `python -m gglasso_amd.build --dev`."""
    load()      # import order (torch first), see above
    if not os.path.exists(DEV_LIB_PATH):
        raise RuntimeError(f"{DEV_LIB_PATH} is missing: python -m gglasso_amd.build --dev")
    return _bind(_bind(ctypes.CDLL(DEV_LIB_PATH), _SIGNATURES), _DEV_SIGNATURES)


def theta_limits():
    """{'GGL': largest K/G of the batched GGL grid, 'FGL': largest K of the FGL Theta-step} (host only, no GPU)."""
    out = (_i * 2)()
    check(load().ggl_theta_limits(out))
    return {"GGL": int(out[0]), "FGL": int(out[1])}


def last_error():
    return load().ggl_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a C return code to the exception the reference would raise (AssertionError for bad
    arguments, solver/admm_solver.py:113-127) or RuntimeError."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == E_ARG:
        raise AssertionError(msg)
    raise RuntimeError(f"libggl_hip error {rc}: {msg}")


def require_gpu():
    n = load().ggl_device_count()
    if n <= 0:
        raise RuntimeError("gglasso_amd needs an AMD GPU (MI355X / gfx950); none is visible "
                           f"(ggl_device_count() = {n}: {last_error()})")
    return n


def as_c(a):
    """C-contiguous float64 view/copy of a NumPy array (the ABI's only data type)."""
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)
