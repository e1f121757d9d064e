"""Builds gglasso_amd/lib/libggl_hip.so (gfx950 only) with hipcc.  In-tree, no JIT cache:
the .so travels to the GPU box with the repo snapshot.

    python -m gglasso_amd.build [--force] [--dev]

--dev additionally builds libggl_hip_dev.so with -DGGL_DEV: the measured-and-rejected product-kernel variants, timing
ablations, probe kernels and the GGL_* environment knobs that tools/ uses.  The solvers never load it.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libggl_hip.so")
SOURCES = ["elementwise.hip", "theta_pair.hip", "ext_group.hip", "eig_jacobi.hip", "recon_gemm.hip", "gemm_sym.hip", "gemm_i8.hip", "deflate.hip", "omega_lds.hip", "newton_schulz.hip", "ggl_comm.hip", "probes_dev.hip",
           # the C ABI, by subject (csrc/capi_internal.hpp has the map)
           "capi_ctx.hip", "capi_omega.hip", "capi_lstep.hip", "capi_batch.hip", "capi_snapshots.hip", "capi_checks.hip",
           "capi_stats.hip", "capi_comm.hip", "capi_ext.hip", "capi_ops.hip"]
HEADERS = ["common.hpp", "kernels.hpp", "ggl_comm.hpp", "capi_internal.hpp", os.path.join("..", "..", "include", "ggl_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs; without it hipcc 7.2 shuttles the f64
# accumulators VGPR<->AGPR around every k-slab (64 extra moves + a full matrix-pipe drain per slab)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


DEV_LIB = os.path.join(LIBDIR, "libggl_hip_dev.so")


def _compile(src, dev=False):
    obj = os.path.join(LIBDIR, src.replace(".hip", ".dev.o" if dev else ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _stale(obj, deps):
        cmd = [HIPCC] + FLAGS + (["-DGGL_DEV"] if dev else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
    return obj


def _link(lib, objs):
    if _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + \
              ["-L/opt/rocm/lib", "-lrocsolver", "-lrocblas", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
        subprocess.check_call(cmd)


def build(force=False, verbose=True, dev=False):
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(LIBDIR):
            if f.endswith(".o") or f.endswith(".so"):
                os.remove(os.path.join(LIBDIR, f))
    with ThreadPoolExecutor(max_workers=7) as ex:
        objs = list(ex.map(_compile, SOURCES))
        dev_objs = list(ex.map(lambda s: _compile(s, True), SOURCES)) if dev else []
    _link(LIB, objs)
    if dev:
        _link(DEV_LIB, dev_objs)
    if verbose:
        print(f"built {LIB}" + (f" and {DEV_LIB}" if dev else ""))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, dev="--dev" in sys.argv)
