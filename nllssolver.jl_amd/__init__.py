"""nllssolver.jl_amd -- MI355X-native Gauss-Newton / Levenberg-Marquardt inner loop for
NLLSsolver.jl (hot path only: residual+Jacobian sweep, block-sparse J'J / J'r accumulation,
damped Schur solve), behind the C ABI of include/nlls_amd.h.

Importing the package never touches the GPU; the first compute call loads
csrc/libnlls_amd.so and FAILS LOUDLY if it is missing or no gfx950 device is visible --
there is no CPU fallback in the product path (oracle/ is test infrastructure only).
"""
from . import kinds
from .kinds import (NoRobust, HuberKernel, Huber2oKernel, GemanMcclureKernel, Scaled)
from .problem import NLLSProblem, runlengthencodesortedints

__all__ = ["kinds", "NLLSProblem", "NoRobust", "HuberKernel", "Huber2oKernel", "GemanMcclureKernel", "Scaled",
           "runlengthencodesortedints"]


def __getattr__(name):
    # lazily expose the device-backed API so that host-only use (tests -m "not gpu") needs no .so
    import importlib
    if name in ("optimize", "optimizesingles", "NLLSOptions", "NLLSResult", "cost", "NLLSIterator", "newton", "levenbergmarquardt", "dogleg",
                "gradientdescent"):
        return getattr(importlib.import_module(__name__ + ".optimizer"), name)
    if name in ("MultiVariateLSgpu", "makesymmvls"):
        return getattr(importlib.import_module(__name__ + ".linearsystem"), name)
    if name in ("nullcallback", "printoutcallback", "storecostscallback", "CostTrajectory"):
        return getattr(importlib.import_module(__name__ + ".callbacks"), name)
    raise AttributeError(name)
