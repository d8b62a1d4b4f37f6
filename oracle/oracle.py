"""ctypes front-end of the CPU oracle (oracle/nlls_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

VARS_CURRENT, VARS_NEXT, VARS_BEST = 0, 1, 2


def build(force=False):
    """Compile liboracle.so with gcc (recipe: oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("nlls_oracle.c", "nlls_oracle.h", "jet.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB


class CostGroup(C.Structure):
    """nlls_cost_group of include/nlls_amd.h."""
    _fields_ = [("res_kind", C.c_int32), ("robust_kind", C.c_int32), ("robust_params", C.c_double * 4),
                ("ncost", C.c_int64), ("varind", C.c_void_p), ("data", C.c_void_p)]


class Info(C.Structure):
    _fields_ = [("is_sparse", C.c_int32), ("has_schur", C.c_int32), ("nvar", C.c_int64), ("nblocks", C.c_int64),
                ("ndof", C.c_int64), ("nnz_data", C.c_int64), ("nblocks_stored", C.c_int64), ("ncost", C.c_int64),
                ("var_storage", C.c_int64), ("nschur_blocks", C.c_int64), ("nreduced_dof", C.c_int64),
                ("owner_path", C.c_int64), ("solve_mode", C.c_int64), ("bandwidth", C.c_int64), ("nborder_dof", C.c_int64)]


class Options(C.Structure):
    _fields_ = [("reldcost", C.c_double), ("absdcost", C.c_double), ("dstep", C.c_double), ("maxfails", C.c_int64),
                ("maxiters", C.c_int64), ("maxtime", C.c_double), ("iterator", C.c_int32), ("store_costs", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("startcost", C.c_double), ("bestcost", C.c_double), ("timetotal", C.c_double), ("timeinit", C.c_double),
                ("timecost", C.c_double), ("timegradient", C.c_double), ("timesolver", C.c_double),
                ("termination", C.c_int64), ("niterations", C.c_int64), ("costcomputations", C.c_int64),
                ("gradientcomputations", C.c_int64), ("linearsolvers", C.c_int64), ("ncosts_stored", C.c_int64),
                ("costs", C.c_double * 512)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p
        L.oracle_robustify.restype = C.c_double
        L.oracle_robustify.argtypes = [C.c_int32, vp, C.c_double]
        L.oracle_robustifydcost.argtypes = [C.c_int32, vp, C.c_double, vp]
        L.oracle_autorobustifydcost.argtypes = [C.c_int32, vp, C.c_double, vp]
        L.oracle_robustifydkernel.argtypes = [vp, C.c_double, vp, vp, vp]
        L.oracle_contaminated_gaussian.argtypes = [C.c_double, C.c_double, C.c_double, vp]
        L.oracle_runlengthencodesortedints.restype = C.c_int64
        L.oracle_runlengthencodesortedints.argtypes = [vp, C.c_int64, vp]
        L.oracle_fast_bAb_dense.restype = C.c_double
        L.oracle_fast_bAb_dense.argtypes = [vp, vp, C.c_int64]
        L.oracle_fast_bAb_csc.restype = C.c_double
        L.oracle_fast_bAb_csc.argtypes = [vp, vp, vp, vp, C.c_int64]
        L.oracle_solve_dense.argtypes = [vp, vp, vp, C.c_int64]
        L.oracle_solve_sparse.argtypes = [vp, vp, vp, vp, vp, C.c_int64]
        L.oracle_var_update.argtypes = [C.c_int32, C.c_int32, vp, vp, vp]
        L.oracle_bsm_build.restype = C.c_int64
        L.oracle_bsm_build.argtypes = [C.c_int64, C.c_int64, vp, vp, vp, vp, vp]
        L.oracle_bsm_to_dense.argtypes = [C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
        L.oracle_bsm_symmetrify_full.argtypes = [C.c_int64, vp, vp, vp, vp, vp, vp]
        L.oracle_bsm_sparse_indices.restype = C.c_int64
        L.oracle_bsm_sparse_indices.argtypes = [C.c_int64, C.c_int64, vp, vp, vp, vp, vp, C.c_int64, C.c_int, vp, vp, vp]
        L.oracle_problem_create.restype = vp
        L.oracle_problem_create.argtypes = [C.c_int64, vp, vp, C.c_int32, vp]
        L.oracle_problem_destroy.argtypes = [vp]
        L.oracle_problem_storage.restype = C.c_int64
        L.oracle_problem_storage.argtypes = [vp]
        L.oracle_set_variables.argtypes = [vp, C.c_int32, vp]
        L.oracle_get_variables.argtypes = [vp, C.c_int32, vp]
        L.oracle_cost.restype = C.c_double
        L.oracle_cost.argtypes = [vp, C.c_int32]
        L.oracle_block_costgradhess.restype = C.c_int
        L.oracle_block_costgradhess.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, vp, vp, vp]
        L.oracle_block_resjac.restype = C.c_int
        L.oracle_block_resjac.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, vp, vp]
        L.oracle_makesymmvls.restype = vp
        L.oracle_makesymmvls.argtypes = [vp, vp, C.c_int32]
        L.oracle_ls_destroy.argtypes = [vp]
        L.oracle_ls_info.argtypes = [vp, vp]
        for f in ("oracle_ls_data", "oracle_ls_b", "oracle_ls_x"):
            getattr(L, f).restype = dp
            getattr(L, f).argtypes = [vp]
        L.oracle_ls_bsm_index.argtypes = [vp, vp, vp, vp, vp]
        L.oracle_costgradhess.restype = C.c_double
        L.oracle_costgradhess.argtypes = [vp, C.c_int32, vp]
        L.oracle_solve_damped.argtypes = [vp, C.c_double]
        L.oracle_max_abs_diag.restype = C.c_double
        L.oracle_max_abs_diag.argtypes = [vp]
        L.oracle_quadform.restype = C.c_double
        L.oracle_quadform.argtypes = [vp, vp, C.c_double]
        L.oracle_update.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
        L.oracle_default_options.argtypes = [vp]
        L.oracle_optimize.argtypes = [vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_groups(groups):
    """groups: list of dicts {res_kind, robust_kind, robust_params, varind (ncost x ndeps int64, 1-based),
    data (ncost x ndata f64)} -> (ctypes array, keepalive list)."""
    arr = (CostGroup * max(len(groups), 1))()
    keep = []
    for i, g in enumerate(groups):
        vi = np.ascontiguousarray(g["varind"], dtype=np.int64)
        da = np.ascontiguousarray(g["data"], dtype=np.float64)
        keep += [vi, da]
        arr[i].res_kind = int(g["res_kind"])
        arr[i].robust_kind = int(g.get("robust_kind", 0))
        rp = list(g.get("robust_params", [])) + [0.0] * 4
        for k in range(4):
            arr[i].robust_params[k] = float(rp[k])
        arr[i].ncost = vi.shape[0]
        arr[i].varind = vi.ctypes.data
        arr[i].data = da.ctypes.data
    return arr, keep


class OracleLS:
    """oracle_ls wrapper: MultiVariateLSsparse / MultiVariateLSdense of src/linearsystem.jl:44-87."""

    def __init__(self, prob, blockindices, flags=0):
        self.prob = prob
        self.bi = np.ascontiguousarray(blockindices, dtype=np.uint64)
        self.h = lib().oracle_makesymmvls(prob.h, _p(self.bi), flags)
        self.info = Info()
        lib().oracle_ls_info(self.h, C.byref(self.info))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_ls_destroy(self.h)
            self.h = None

    @property
    def data(self):
        return np.ctypeslib.as_array(lib().oracle_ls_data(self.h), shape=(self.info.nnz_data,))

    @property
    def b(self):
        return np.ctypeslib.as_array(lib().oracle_ls_b(self.h), shape=(self.info.ndof,))

    @property
    def x(self):
        return np.ctypeslib.as_array(lib().oracle_ls_x(self.h), shape=(self.info.ndof,))

    def bsm_index(self):
        nb, ns = self.info.nblocks, self.info.nblocks_stored
        cp = np.zeros(nb + 1, np.int64); rv = np.zeros(max(ns, 1), np.int64); nz = np.zeros(max(ns, 1), np.int64)
        bo = np.zeros(nb, np.int64)
        lib().oracle_ls_bsm_index(self.h, _p(cp), _p(rv), _p(nz), _p(bo))
        return cp, rv[:ns], nz[:ns], bo

    def costgradhess(self, which=VARS_CURRENT):
        return lib().oracle_costgradhess(self.prob.h, which, self.h)

    def solve(self, lam=0.0):
        return lib().oracle_solve_damped(self.h, lam)

    def max_abs_diag(self):
        return lib().oracle_max_abs_diag(self.h)

    def quadform(self, x, lam=0.0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().oracle_quadform(self.h, _p(x), lam)


class OracleProblem:
    """oracle_problem wrapper: the NLLSProblem of src/problem.jl:5-20 in packed form."""

    def __init__(self, var_kind, var_dim, groups):
        self.var_kind = np.ascontiguousarray(var_kind, dtype=np.int32)
        self.var_dim = np.ascontiguousarray(var_dim, dtype=np.int32)
        self.garr, self._keep = make_groups(groups)
        self.ngroups = len(groups)
        self.h = lib().oracle_problem_create(len(self.var_kind), _p(self.var_kind), _p(self.var_dim), self.ngroups, self.garr)
        self.nstorage = lib().oracle_problem_storage(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_problem_destroy(self.h)
            self.h = None

    def set_variables(self, packed, which=VARS_CURRENT):
        packed = np.ascontiguousarray(packed, dtype=np.float64)
        assert packed.size == self.nstorage
        lib().oracle_set_variables(self.h, which, _p(packed))

    def get_variables(self, which=VARS_CURRENT):
        out = np.zeros(self.nstorage)
        lib().oracle_get_variables(self.h, which, _p(out))
        return out

    def cost(self, which=VARS_CURRENT):
        return lib().oracle_cost(self.h, which)

    def block_costgradhess(self, gi, ci, which=VARS_CURRENT):
        c = C.c_double(); g = np.zeros(16); H = np.zeros(256)
        P = lib().oracle_block_costgradhess(self.h, which, gi, ci, C.byref(c), _p(g), _p(H))
        return c.value, g[:P].copy(), H[:P * P].reshape(P, P).T.copy()

    def block_resjac(self, gi, ci, M, which=VARS_CURRENT):
        r = np.zeros(4); J = np.zeros(64)
        n = lib().oracle_block_resjac(self.h, which, gi, ci, _p(r), _p(J))
        return r[:M].copy(), J[:M * n].reshape(n, M).T.copy()

    def linear_system(self, blockindices=None, flags=0):
        if blockindices is None:
            blockindices = np.arange(1, len(self.var_kind) + 1, dtype=np.uint64)
        return OracleLS(self, blockindices, flags)

    def update(self, ls, to=VARS_NEXT, frm=VARS_CURRENT, step=None):
        s = None if step is None else np.ascontiguousarray(step, dtype=np.float64)
        lib().oracle_update(self.h, to, frm, ls.h, _p(s))

    def optimize(self, blockindices=None, **kw):
        if blockindices is None:
            blockindices = np.arange(1, len(self.var_kind) + 1, dtype=np.uint64)
        bi = np.ascontiguousarray(blockindices, dtype=np.uint64)
        opt = Options(); lib().oracle_default_options(C.byref(opt))
        for k, v in kw.items():
            setattr(opt, k, v)
        res = Result()
        lib().oracle_optimize(self.h, _p(bi), C.byref(opt), C.byref(res))
        return res
