"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol include/nlls_amd.h
declares, the static helpers agree with the host registry, and the product path FAILS LOUDLY without a GPU."""
import os
import re

import numpy as np
import pytest

from nllssolver_jl_amd import _capi, kinds as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    h = open(os.path.join(ROOT, "include", "nlls_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(nlls_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    L = _capi.lib()
    names = header_functions()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"libnlls_amd.so lacks {missing}"
    assert sorted(_capi.SYMBOLS) == names, set(_capi.SYMBOLS) ^ set(names)


def test_static_helpers_match_registry():
    L = _capi.lib()
    for kind, (ndeps, nres, ndata, adaptive, slots) in K.RES_TABLE.items():
        assert L.nlls_res_ndeps(kind) == ndeps and L.nlls_res_nres(kind) == nres and L.nlls_res_ndata(kind) == ndata
        for s, (vk, vd) in enumerate(slots):
            a, b = np.zeros(1, np.int32), np.zeros(1, np.int32)
            assert L.nlls_res_slot_kind(kind, s, _capi._p(a), _capi._p(b)) == 0
            assert a[0] == vk and (vk != K.VAR_EUCLIDEAN or b[0] == vd)
    assert L.nlls_res_ndeps(999) == _capi.ERR_UNSUPPORTED
    for kind, dim in ((K.VAR_EUCLIDEAN, 6), (K.VAR_ZERO_TO_INF, 1), (K.VAR_ZERO_TO_ONE, 1), (K.VAR_CONTAMINATED_GAUSSIAN, 3), (K.VAR_POSE_SO3, 6)):
        assert L.nlls_var_storage(kind, dim) == K.var_storage(kind, dim) and L.nlls_var_dof(kind, dim) == K.var_dof(kind, dim)


def test_entry_points_never_throw_across_the_boundary():
    """nothing throws or longjmps across the ccall boundary.  Every extern "C" body of csrc/nlls_capi.cpp, nlls_lm.cpp and nlls_comm.cpp sits between NLLS_API_BEGIN /
    NLLS_API_END (a C++ exception becomes NLLS_ERR_HIP + nlls_last_error), and a graph size no reduced system has is refused before a work vector is sized by it: an error code
    comes back, not std::terminate."""
    import ctypes as C
    L = _capi.lib()
    n = 2 ** 31 - 1
    adjptr = np.zeros(2, np.int64); adj = np.zeros(1, np.int32); perm = np.zeros(1, np.int32)
    assert L.nlls_rcm_order(n, _capi._p(adjptr), _capi._p(adj), _capi._p(perm)) == _capi.ERR_INVALID_ARG
    one = np.zeros(4, np.int32); col = np.zeros(4, np.int64)
    assert L.nlls_nd_tiles(n, 0, _capi._p(adjptr), _capi._p(adj), _capi._p(one), _capi._p(one), _capi._p(one), 1, _capi._p(one), _capi._p(one), _capi._p(col), 0, None) == _capi.ERR_INVALID_ARG
    assert L.nlls_rcm_order(-1, None, None, None) == _capi.ERR_INVALID_ARG
    # the guard pair is on every entry point that returns a status (a source check: the macro pair cannot be observed from outside without exhausting the host's memory)
    import re
    for f in ("nlls_capi.cpp", "nlls_lm.cpp", "nlls_comm.cpp"):
        src = open(os.path.join(ROOT, "nllssolver.jl_amd", "csrc", f)).read()
        heads = re.findall(r"^(?:extern \"C\" )?int\s+(nlls_\w+)\([^\n]*\)\s*\{([^\n]*)$", src, flags=re.M)
        assert heads, f
        for name, rest in heads:
            assert "NLLS_API_BEGIN" in rest or name == "nlls_upload_structure", (f, name)      # (nlls_upload_structure carries its try / catch written out)
    assert src.count("NLLS_API_BEGIN") >= 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(_capi.NllsError) as e:
        _capi.Context()
    assert e.value.code == _capi.ERR_NO_DEVICE
    import nllssolver_jl_amd as N
    p = N.NLLSProblem(); p.addvariable(0.0); p.addcosts(K.RES_ROSENBROCK_A, [[1]], [[1.0]])
    with pytest.raises(_capi.NllsError):
        N.cost(p)
    with pytest.raises(_capi.NllsError):
        N.optimize(p)


def test_host_problem_container():
    import nllssolver_jl_amd as N
    p = N.NLLSProblem()
    assert p.addvariable([1.0, 2, 3, 4, 5, 6]) == 1 and p.addvariable([0.0, 0, 1]) == 2
    p.addcosts(K.RES_BA_AFFINE, [[1, 2]], [[0.5, 0.25]], N.HuberKernel(1.0))
    assert p.ncosts() == 1 and p.nresiduals() == 2
    with pytest.raises(AssertionError):
        p.addcosts(K.RES_BA_AFFINE, [[2, 1]], [[0.0, 0.0]])          # slot kinds do not match
    with pytest.raises(AssertionError):
        p.addcosts(K.RES_BA_AFFINE, [[1, 3]], [[0.0, 0.0]])          # variable index out of range
    (g,) = p.groups()
    assert g["robust_kind"] == K.ROBUST_HUBER and g["varind"].dtype == np.int64 and g["varind"].min() == 1
