"""gglasso_amd -- the ADMM inner loop of General Graphical Lasso (fabian-sp/GGLasso) as gfx950 HIP
kernels behind a C ABI, with drop-in ``ADMM_MGL`` / ``ADMM_SGL`` solvers and the reference's operator
names.  There is no CPU path in this package."""
from .solver import ADMM_MGL, ADMM_SGL, block_SGL  # noqa: F401
from .ext_solver import ext_ADMM_MGL  # noqa: F401

__all__ = ["ADMM_MGL", "ADMM_SGL", "block_SGL", "ext_ADMM_MGL"]
