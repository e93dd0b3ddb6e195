"""cholesky_amd -- MI355X-native supernodal sparse Cholesky hot path (drop-in for the per-supernode
POTRF/TRSM/SYRK/GEMM path of syamajala/cholesky, mmat.rg:1227-1355 -> blas.rg:292-504).

The product is the C-ABI shared library (include/cholamd.h, cholesky_amd/lib/libcholamd.so): C host
code + hand-written HIP kernels for gfx950.  This package is a thin host-side mirror used by the
tests, bench.py and the multi-GPU driver; torch is used only for device memory, streams and
torch.distributed (RCCL).  Nothing here computes on the CPU.
"""
from ._lib import CholamdError, Filled, Op, Region, load  # noqa: F401
from .plan import Plan, Problem  # noqa: F401
from .device import Comm, Device  # noqa: F401
from . import blas  # noqa: F401

__all__ = ["Plan", "Problem", "Device", "Comm", "blas", "CholamdError", "Filled", "Op", "Region", "load"]
