"""Builds gglasso_amd/lib/libggl_hip.so (gfx950 only) with hipcc.  In-tree, no JIT cache:
the .so travels to the GPU box with the repo snapshot.

    python -m gglasso_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libggl_hip.so")
SOURCES = ["elementwise.hip", "theta_pair.hip", "eig_jacobi.hip", "recon_gemm.hip", "gemm_sym.hip", "newton_schulz.hip", "ggl_capi.hip"]
HEADERS = ["common.hpp", "kernels.hpp", os.path.join("..", "..", "include", "ggl_hip.h")]
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


def _compile(src):
    obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _stale(obj, deps):
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(LIBDIR):
            if f.endswith(".o") or f.endswith(".so"):
                os.remove(os.path.join(LIBDIR, f))
    with ThreadPoolExecutor(max_workers=7) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
              ["-L/opt/rocm/lib", "-lrocsolver", "-lrocblas", "-Wl,-rpath,/opt/rocm/lib"]
        subprocess.check_call(cmd)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
