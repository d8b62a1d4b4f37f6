"""USER residual kinds (include/nlls_amd.h, NLLS_RES_USER0 .. 7): the reference's open world -- any Julia function as a residual, differentiated by ForwardDiff
(src/autodiff.jl:81-93) -- reaches the device at BUILD time: a user header with ONE templated eval<T> per kind, `make user USER_KINDS=...`, and every kernel of the path is
instantiated for it.  __graft_entry__.build() builds the example library (tests/user_kinds/radial_ba.hpp -> csrc/libnlls_amd_userdemo.so); the checks run in a process of
their own (tests/userkind_worker.py), because the library is chosen by NLLS_AMD_LIB before it is loaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USERLIB = os.path.join(ROOT, "nllssolver.jl_amd", "csrc", "libnlls_amd_userdemo.so")


def test_user_library_exports_the_same_abi():
    """(CPU) the library with user kinds is the same C ABI: every symbol of include/nlls_amd.h"""
    import ctypes
    from nllssolver_jl_amd import _capi
    assert os.path.exists(USERLIB), "run __graft_entry__.build()"
    L = ctypes.CDLL(USERLIB)
    assert not [s for s in _capi.SYMBOLS if not hasattr(L, s)]


@pytest.mark.gpu
def test_user_kinds_on_the_device():
    env = dict(os.environ, NLLS_AMD_LIB=USERLIB)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "userkind_worker.py")], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "user kinds ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
