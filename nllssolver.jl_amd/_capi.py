"""ctypes binding of csrc/libnlls_amd.so (the C ABI of include/nlls_amd.h).

The product path has NO CPU fallback: if the shared library is missing, or no gfx950 device is
visible, every compute entry point raises NllsError.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NLLS_AMD_LIB: another build of the library -- one with USER residual kinds (include/nlls_amd.h, NLLS_RES_USER0 .. 7; the Julia shim reads the same variable)
LIB_PATH = os.environ.get("NLLS_AMD_LIB") or os.path.join(_HERE, "csrc", "libnlls_amd.so")

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NOT_READY, ERR_NOT_SPD, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6
FLAG_FORCE_ATOMIC, FLAG_NO_SCHUR, FLAG_FORCE_SPARSE, FLAG_NO_BAND, FLAG_NO_TWIST, FLAG_NO_BCR, FLAG_DETERMINISTIC = 1, 2, 4, 8, 16, 32, 64
FLAG_PRESHARDED = 128
FLAG_NO_REORDER = 256
FLAG_NO_TILE_SPARSE = 512
FLAG_NO_PIVOT_FLOOR = 1024
FLAG_MATERIALIZE = 2048
OPT_MATERIALIZE, OPT_LOOKAHEAD, OPT_PHASE_EVENTS = 1, 2, 3
VARS_CURRENT, VARS_NEXT, VARS_BEST = 0, 1, 2

# every symbol include/nlls_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = """nlls_ctx_create nlls_ctx_destroy nlls_last_error nlls_set_stream nlls_set_shard nlls_var_storage nlls_var_dof
nlls_res_ndeps nlls_res_nres nlls_res_ndata nlls_res_slot_kind nlls_rcm_order nlls_nd_tiles nlls_upload_structure nlls_get_info nlls_get_bsm_index
nlls_set_variables nlls_get_variables nlls_swap_variables nlls_copy_variables nlls_sweep_gradhess nlls_sweep_cost
nlls_get_grad nlls_get_bsm_data nlls_max_abs_diag nlls_grad_sqnorm nlls_grad_quadform nlls_damp nlls_solve nlls_get_solve_stats nlls_set_step
nlls_get_step nlls_step_maxabs nlls_step_norm nlls_quadform nlls_retract nlls_sweep_gradhess_local
nlls_sweep_gradhess_finish nlls_sweep_cost_local nlls_sweep_cost_finish nlls_solve_local nlls_solve_finish
nlls_get_reduce_buffer nlls_get_step_shard nlls_get_shard_info nlls_get_grad_owned nlls_trial_local nlls_solve_finish_async nlls_lm_trial nlls_optimize_singles nlls_time_sweep_gradhess nlls_time_sweep_accumulate nlls_time_sweep_cost nlls_time_solve nlls_time_reduced_solve nlls_profile_sweep nlls_profile_sweep_dispatch nlls_solve_finish_replicated nlls_get_variables_owned nlls_lm_iterations
nlls_set_allreduce nlls_comm_unique_id nlls_comm_init_rccl nlls_comm_post_flag nlls_comm_agreed_flag nlls_comm_info nlls_get_memory_info nlls_flush_cache nlls_check_analytic nlls_set_option nlls_get_time_buckets nlls_get_phase_times""".split()


class LmOptions(C.Structure):          # nlls_lm_options
    _fields_ = [("reldcost", C.c_double), ("absdcost", C.c_double), ("dstep", C.c_double), ("maxfails", C.c_int64), ("maxiters", C.c_int64),
                ("stoptime_ns", C.c_int64)]


class LmState(C.Structure):            # nlls_lm_state
    _fields_ = [("lambda_", C.c_double), ("bestcost", C.c_double), ("cost", C.c_double), ("iternum", C.c_int64), ("fails", C.c_int64),
                ("have_best", C.c_int64), ("converged", C.c_int64), ("linearsolvers", C.c_int64), ("costcomputations", C.c_int64),
                ("gradientcomputations", C.c_int64), ("singulartrials", C.c_int64), ("timesolver_ns", C.c_int64), ("timegradient_ns", C.c_int64), ("timecost_ns", C.c_int64)]


class NllsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nlls_amd error {code}: {msg}")
        self.code = code


class CostGroup(C.Structure):
    _fields_ = [("res_kind", C.c_int32), ("robust_kind", C.c_int32), ("robust_params", C.c_double * 4),
                ("ncost", C.c_int64), ("varind", C.c_void_p), ("data", C.c_void_p)]


class Info(C.Structure):
    _fields_ = [("is_sparse", C.c_int32), ("has_schur", C.c_int32), ("nvar", C.c_int64), ("nblocks", C.c_int64),
                ("ndof", C.c_int64), ("nnz_data", C.c_int64), ("nblocks_stored", C.c_int64), ("ncost", C.c_int64),
                ("var_storage", C.c_int64), ("nschur_blocks", C.c_int64), ("nreduced_dof", C.c_int64),
                ("owner_path", C.c_int64), ("solve_mode", C.c_int64), ("bandwidth", C.c_int64), ("nborder_dof", C.c_int64)]


def build(force=False):
    """Compile libnlls_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", csrc, "libnlls_amd.so"])
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library or fail loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NllsError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C nllssolver.jl_amd/csrc); "
                                           "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.nlls_ctx_create.argtypes = [vp, i32, C.POINTER(vp)]
        L.nlls_ctx_destroy.argtypes = [vp]
        L.nlls_last_error.argtypes = [vp]; L.nlls_last_error.restype = C.c_char_p
        L.nlls_set_stream.argtypes = [vp, vp]
        L.nlls_set_shard.argtypes = [vp, i32, i32]
        L.nlls_var_storage.argtypes = [i32, i32]; L.nlls_var_dof.argtypes = [i32, i32]
        L.nlls_res_ndeps.argtypes = [i32]; L.nlls_res_nres.argtypes = [i32]; L.nlls_res_ndata.argtypes = [i32]
        L.nlls_res_slot_kind.argtypes = [i32, i32, vp, vp]
        L.nlls_rcm_order.argtypes = [i32, vp, vp, vp]
        L.nlls_nd_tiles.argtypes = [i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i64, vp]
        L.nlls_upload_structure.argtypes = [vp, i64, vp, vp, vp, i32, vp, i32]
        L.nlls_get_info.argtypes = [vp, vp]
        L.nlls_lm_iterations.argtypes = [vp, vp, vp, i64]
        L.nlls_get_bsm_index.argtypes = [vp, vp, vp, vp, vp]
        L.nlls_set_variables.argtypes = [vp, i32, vp]; L.nlls_get_variables.argtypes = [vp, i32, vp]
        L.nlls_swap_variables.argtypes = [vp, i32, i32]; L.nlls_copy_variables.argtypes = [vp, i32, i32]
        L.nlls_sweep_gradhess.argtypes = [vp, vp]; L.nlls_sweep_cost.argtypes = [vp, i32, vp]
        L.nlls_get_grad.argtypes = [vp, vp]; L.nlls_get_bsm_data.argtypes = [vp, vp]; L.nlls_get_grad_owned.argtypes = [vp, vp]
        L.nlls_max_abs_diag.argtypes = [vp, vp]; L.nlls_grad_sqnorm.argtypes = [vp, vp]; L.nlls_grad_quadform.argtypes = [vp, vp]
        L.nlls_damp.argtypes = [vp, dbl]
        L.nlls_get_solve_stats.argtypes = [vp, vp, i32]
        L.nlls_solve.argtypes = [vp, vp]; L.nlls_set_step.argtypes = [vp, vp]; L.nlls_get_step.argtypes = [vp, vp]
        L.nlls_step_maxabs.argtypes = [vp, vp]; L.nlls_step_norm.argtypes = [vp, vp]
        L.nlls_quadform.argtypes = [vp, vp, vp]
        L.nlls_retract.argtypes = [vp, i32, i32]; L.nlls_lm_trial.argtypes = [vp, C.c_double, i32, i32, vp]
        L.nlls_optimize_singles.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, i32, i32, i32, C.c_double, C.c_double, C.c_double, vp]
        L.nlls_sweep_gradhess_local.argtypes = [vp]; L.nlls_sweep_gradhess_finish.argtypes = [vp, vp]
        L.nlls_sweep_cost_local.argtypes = [vp, i32]; L.nlls_sweep_cost_finish.argtypes = [vp, vp]
        L.nlls_solve_local.argtypes = [vp]; L.nlls_solve_finish.argtypes = [vp, vp]
        L.nlls_get_reduce_buffer.argtypes = [vp, i32, vp, vp]; L.nlls_trial_local.argtypes = [vp, i32, i32, vp]; L.nlls_solve_finish_async.argtypes = [vp]
        L.nlls_get_step_shard.argtypes = [vp, vp, vp, vp, vp]
        L.nlls_get_shard_info.argtypes = [vp, vp, i32]
        L.nlls_time_sweep_gradhess.argtypes = [vp, i32, vp]; L.nlls_time_sweep_cost.argtypes = [vp, i32, vp]; L.nlls_time_sweep_accumulate.argtypes = [vp, i32, vp]
        L.nlls_time_solve.argtypes = [vp, i32, vp]; L.nlls_time_reduced_solve.argtypes = [vp, i32, vp]
        L.nlls_profile_sweep.argtypes = [vp, i32, vp, vp, vp, vp]; L.nlls_profile_sweep_dispatch.argtypes = [vp, vp, vp, vp, vp]
        L.nlls_solve_finish_replicated.argtypes = [vp]; L.nlls_get_variables_owned.argtypes = [vp, i32, vp]
        L.nlls_set_allreduce.argtypes = [vp, vp, vp]; L.nlls_comm_unique_id.argtypes = [vp]; L.nlls_comm_init_rccl.argtypes = [vp, vp]
        L.nlls_get_memory_info.argtypes = [vp, vp, i32]; L.nlls_flush_cache.argtypes = [vp, i64]; L.nlls_check_analytic.argtypes = [vp, vp, i32]
        L.nlls_set_option.argtypes = [vp, i32, i64]; L.nlls_get_time_buckets.argtypes = [vp, vp, i32]; L.nlls_get_phase_times.argtypes = [vp, vp, i32]
        L.nlls_comm_post_flag.argtypes = [vp, dbl]; L.nlls_comm_agreed_flag.argtypes = [vp, dbl, vp]; L.nlls_comm_info.argtypes = [vp, vp, i32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """Owns one nlls_ctx."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        dev = np.array([device], np.int32)
        rc = self.L.nlls_ctx_create(_p(dev), 1, C.byref(self.h))
        if rc != OK:
            self.h = None
            raise NllsError(rc, "nlls_ctx_create failed: no gfx950 (MI355X) device visible -- there is no CPU fallback"
                            if rc == ERR_NO_DEVICE else "nlls_ctx_create failed")
        self.info = None
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.nlls_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            exc = getattr(self, "_reduce_exc", None)
            if exc is not None:             # the installed all-reduce raised inside the library's call: that exception is the cause, not "all-reduce failed"
                self._reduce_exc = None
                raise NllsError(rc, self.L.nlls_last_error(self.h).decode() + f" ({type(exc).__name__}: {exc})") from exc
            raise NllsError(rc, self.L.nlls_last_error(self.h).decode())
        return rc

    def set_stream(self, stream_ptr):
        self._chk(self.L.nlls_set_stream(self.h, C.c_void_p(stream_ptr)))

    def upload(self, var_kind, var_dim, blockindices, groups, flags=0):
        """nlls_upload_structure; groups = list of dicts as produced by NLLSProblem.groups()."""
        vk = np.ascontiguousarray(var_kind, np.int32); vd = np.ascontiguousarray(var_dim, np.int32)
        bi = np.ascontiguousarray(blockindices, np.uint64)
        arr = (CostGroup * max(len(groups), 1))()
        keep = [vk, vd, bi]
        for i, g in enumerate(groups):
            vi = np.ascontiguousarray(g["varind"], np.int64); da = np.ascontiguousarray(g["data"], np.float64)
            keep += [vi, da]
            arr[i].res_kind = int(g["res_kind"]); arr[i].robust_kind = int(g.get("robust_kind", 0))
            rp = list(g.get("robust_params", ())) + [0.0] * 4
            for k in range(4):
                arr[i].robust_params[k] = float(rp[k])
            arr[i].ncost = vi.shape[0]; arr[i].varind = vi.ctypes.data; arr[i].data = da.ctypes.data
        self._chk(self.L.nlls_upload_structure(self.h, len(vk), _p(vk), _p(vd), _p(bi), len(groups), arr, flags))
        self.info = Info()
        self._chk(self.L.nlls_get_info(self.h, C.byref(self.info)))
        return self.info

    def bsm_index(self):
        nb, ns = self.info.nblocks, self.info.nblocks_stored
        cp = np.zeros(nb + 1, np.int64); rv = np.zeros(max(ns, 1), np.int64); nz = np.zeros(max(ns, 1), np.int64)
        bo = np.zeros(max(nb, 1), np.int64)
        self._chk(self.L.nlls_get_bsm_index(self.h, _p(cp), _p(rv), _p(nz), _p(bo)))
        return cp, rv[:ns], nz[:ns], bo[:nb]

    def set_variables(self, packed, which=VARS_CURRENT):
        packed = np.ascontiguousarray(packed, np.float64)
        assert packed.size == self.info.var_storage
        self._chk(self.L.nlls_set_variables(self.h, which, _p(packed)))

    def get_variables(self, which=VARS_CURRENT):
        out = np.zeros(self.info.var_storage)
        self._chk(self.L.nlls_get_variables(self.h, which, _p(out)))
        return out

    def swap_variables(self, a, b):
        self._chk(self.L.nlls_swap_variables(self.h, a, b))

    def copy_variables(self, dst, src):
        self._chk(self.L.nlls_copy_variables(self.h, dst, src))

    def _scalar(self, fn, *args):
        out = C.c_double()
        self._chk(fn(self.h, *args, C.byref(out)))
        return out.value

    def sweep_gradhess(self, want_cost=True):
        if not want_cost:                          # enqueue only: no cost reduction, no synchronisation
            self._chk(self.L.nlls_sweep_gradhess(self.h, None)); return None
        return self._scalar(self.L.nlls_sweep_gradhess)

    def sweep_cost(self, which=VARS_CURRENT):
        return self._scalar(self.L.nlls_sweep_cost, which)

    def get_grad(self):
        out = np.zeros(self.info.ndof); self._chk(self.L.nlls_get_grad(self.h, _p(out))); return out

    def get_grad_owned(self):
        out = np.zeros(self.info.ndof); self._chk(self.L.nlls_get_grad_owned(self.h, _p(out))); return out

    def get_bsm_data(self):
        out = np.zeros(self.info.nnz_data); self._chk(self.L.nlls_get_bsm_data(self.h, _p(out))); return out

    def max_abs_diag(self):
        return self._scalar(self.L.nlls_max_abs_diag)

    def grad_sqnorm(self):
        return self._scalar(self.L.nlls_grad_sqnorm)

    def grad_quadform(self):
        return self._scalar(self.L.nlls_grad_quadform)

    def damp(self, delta):
        self._chk(self.L.nlls_damp(self.h, float(delta)))

    def solve(self, want_x=False):
        out = np.zeros(self.info.ndof) if want_x else None
        self._chk(self.L.nlls_solve(self.h, _p(out)))
        return out

    def lm_trial(self, dlambda, to=VARS_NEXT, frm=VARS_CURRENT):
        """damp + solve + retract + cost sweep in one call / one synchronisation; returns the trial cost."""
        out = C.c_double()
        self._chk(self.L.nlls_lm_trial(self.h, float(dlambda), to, frm, C.byref(out)))
        return out.value

    def optimize_singles(self, varindices, cptr, cgroup, cindex, cslot, maxiters=100, maxfails=3, reldcost=1e-15, absdcost=1e-15, dstep=1e-15, iterator=1):
        """nlls_optimize_singles; returns the iterations each listed variable took."""
        vi = np.ascontiguousarray(varindices, np.int64); cp = np.ascontiguousarray(cptr, np.int64)
        cg = np.ascontiguousarray(cgroup, np.int32); ci = np.ascontiguousarray(cindex, np.int64); cs = np.ascontiguousarray(cslot, np.int32)
        assert cp.size == vi.size + 1 and cg.size == ci.size == cs.size == (int(cp[-1]) if cp.size else 0)
        iters = np.zeros(vi.size, np.int64)
        self._chk(self.L.nlls_optimize_singles(self.h, vi.size, _p(vi), _p(cp), _p(cg), _p(ci), _p(cs), int(iterator), int(maxiters), int(maxfails),
                                                float(reldcost), float(absdcost), float(dstep), _p(iters)))
        return iters

    # ---- collectives behind the ABI (include/nlls_amd.h) ------------------------------------------------------
    ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)

    def set_allreduce(self, fn):
        """fn(dev_ptr, count, op, hip_stream) -> 0: in-place all-reduce of `count` doubles in device memory (op 0 sum, 1 max).  The LM
        loop's entry points then run collectively inside the library (nlls_lm_iterations included).  fn = None removes it."""
        if fn is None:
            self._reduce_cb = None
            self._chk(self.L.nlls_set_allreduce(self.h, None, None)); return
        def tramp(user, ptr, count, op, stream):
            try:
                return int(fn(ptr, count, op, stream) or 0)
            except Exception as e:          # (an exception must not unwind through the C frames)
                self._reduce_exc = e; return 1
        self._reduce_cb = self.ALLREDUCE_FN(tramp)       # kept alive with the context
        self._chk(self.L.nlls_set_allreduce(self.h, self._reduce_cb, None))

    @staticmethod
    def comm_unique_id():
        buf = (C.c_ubyte * 128)()
        rc = lib().nlls_comm_unique_id(buf)
        if rc != OK:
            raise NllsError(rc, "nlls_comm_unique_id: librccl could not be loaded")
        return bytes(buf)

    def comm_init_rccl(self, id128):
        """RCCL inside the library: one communicator for this context (after set_shard), collectives on the context's stream."""
        buf = (C.c_ubyte * 128).from_buffer_copy(id128)
        self._chk(self.L.nlls_comm_init_rccl(self.h, buf))

    def memory_info(self):
        out = np.zeros(5, np.int64); self._chk(self.L.nlls_get_memory_info(self.h, _p(out), 5))
        return dict(working_set_bytes=int(out[0]), arena_bytes=int(out[1]), a_data_bytes=int(out[2]), reduced_system_bytes=int(out[3]), sharded_reduce_bytes=int(out[4]))

    def check_analytic(self):
        """closed-form Jacobians / kernel derivatives against the dual-number statement, per quantity (include/nlls_amd.h)"""
        out = np.zeros(7); self._chk(self.L.nlls_check_analytic(self.h, _p(out), 7))
        return dict(zip(("J", "Jtr", "cost", "drho", "d2rho", "dkernel", "d2kernel"), out.tolist()))

    def flush_cache(self, nbytes):
        self._chk(self.L.nlls_flush_cache(self.h, int(nbytes)))

    def comm_post_flag(self, value):
        self._chk(self.L.nlls_comm_post_flag(self.h, float(value)))

    def comm_agreed_flag(self, local_value):
        out = C.c_double(0.0); self._chk(self.L.nlls_comm_agreed_flag(self.h, float(local_value), C.byref(out))); return float(out.value)

    def comm_info(self):
        """what the library's own communicator reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), not the launcher's environment"""
        out = np.zeros(4, np.int64); self._chk(self.L.nlls_comm_info(self.h, _p(out), 4))
        return dict(nranks=int(out[0]), rank=int(out[1]), device=int(out[2]), transport={0: "none", 1: "rccl", 2: "caller-installed all-reduce"}[int(out[3])])

    def set_option(self, option, value):
        """nlls_set_option: OPT_MATERIALIZE (1: nlls_lm_trial eliminates from the materialised A.data, 0: matrix-free where it applies), OPT_LOOKAHEAD"""
        self._chk(self.L.nlls_set_option(self.h, int(option), int(value)))

    def phase_times(self):
        """nlls_get_phase_times: microseconds per collective trial by phase (stream events; NLLS_OPT_PHASE_EVENTS) and per gradient sweep"""
        out = np.zeros(8); self._chk(self.L.nlls_get_phase_times(self.h, _p(out), 8)); nt, ns = max(out[6], 1.0), max(out[7], 1.0)
        return dict(trials=int(out[6]), sweeps=int(out[7]), elimination_us=1e3 * out[0] / nt, allreduce_S_us=1e3 * out[1] / nt, reduced_solve_us=1e3 * out[2] / nt,
                    backsub_retraction_us=1e3 * out[3] / nt, trial_tail_us=1e3 * out[4] / nt, gradient_sweep_us=1e3 * out[5] / ns)

    def time_buckets(self):
        """device-timed NLLSResult buckets since the upload: seconds (gradient, cost, solver) and the trials counted"""
        out = np.zeros(4, np.int64); self._chk(self.L.nlls_get_time_buckets(self.h, _p(out), 4))
        return dict(timegradient=out[0] * 1e-9, timecost=out[1] * 1e-9, timesolver=out[2] * 1e-9, trials=int(out[3]))

    def solve_stats(self):
        out = np.zeros(27, np.int64); self._chk(self.L.nlls_get_solve_stats(self.h, _p(out), 27))
        return dict(status=int(out[0]), band_factor_cycles=int(out[1]), band_backward_cycles=int(out[2]), solve_mode=int(out[3]),
                    elim_supernodes=int(out[4]), bandwidth=int(out[5]), bcr_mfma_issued=int(out[6]), bcr_launches=int(out[7]), bcr_levels=int(out[8]),
                    band_dof=int(out[9]), dropped_pivots=int(out[10]), reduced_row_sums=int(out[11]), lazy_trials=int(out[12]),
                    reordered=int(out[13]), bandwidth_caller_order=int(out[14]), dense_window=int(out[15]),
                    tsp_tiles=int(out[16]), tsp_levels=int(out[17]), tsp_lower_tiles=int(out[18]), tsp_launches=int(out[19]), tsp_products=int(out[20]),
                    lookahead_hits=int(out[21]), lookahead_misses=int(out[22]), mf_trials=int(out[23]), reduced_sweeps=int(out[24]), full_sweeps=int(out[25]), bcr_block=int(out[26]))

    def set_step(self, x):
        x = np.ascontiguousarray(x, np.float64); assert x.size == self.info.ndof
        self._chk(self.L.nlls_set_step(self.h, _p(x)))

    def get_step(self):
        out = np.zeros(self.info.ndof); self._chk(self.L.nlls_get_step(self.h, _p(out))); return out

    def step_maxabs(self):
        return self._scalar(self.L.nlls_step_maxabs)

    def step_norm(self):
        return self._scalar(self.L.nlls_step_norm)

    def quadform(self):
        a, b = C.c_double(), C.c_double()
        self._chk(self.L.nlls_quadform(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def retract(self, to=VARS_NEXT, frm=VARS_CURRENT):
        self._chk(self.L.nlls_retract(self.h, to, frm))

    # ---- sharding ------------------------------------------------------------------------------------------
    def set_shard(self, rank, nranks):
        self._chk(self.L.nlls_set_shard(self.h, rank, nranks))

    def shard_info(self):
        out = np.zeros(6, np.int64); self._chk(self.L.nlls_get_shard_info(self.h, _p(out), 6))
        return dict(rank=int(out[0]), nranks=int(out[1]), local_ncost=int(out[2]), local_nnz_data=int(out[3]), local_ndof=int(out[4]), replicated=int(out[5]))

    def sweep_gradhess_local(self):
        self._chk(self.L.nlls_sweep_gradhess_local(self.h))

    def solve_finish_async(self):
        self._chk(self.L.nlls_solve_finish_async(self.h))

    def trial_local(self, to=VARS_NEXT, frm=VARS_CURRENT):
        out = np.zeros(6); self._chk(self.L.nlls_trial_local(self.h, to, frm, _p(out))); return out

    def lm_iterations(self, options, state, niter):
        """nlls_lm_iterations: up to niter outer Levenberg-Marquardt iterations in the library's own host loop (state updated in place)"""
        self._chk(self.L.nlls_lm_iterations(self.h, C.byref(options), C.byref(state), int(niter)))

    def trial_local_enqueue(self, to=VARS_NEXT, frm=VARS_CURRENT):
        """no synchronisation: the scalars stay on the device (reduce_buffer(3))"""
        self._chk(self.L.nlls_trial_local(self.h, to, frm, None))

    def solve_finish_replicated(self):
        self._chk(self.L.nlls_solve_finish_replicated(self.h))

    def get_variables_owned(self, which=VARS_CURRENT):
        out = np.zeros(self.info.var_storage); self._chk(self.L.nlls_get_variables_owned(self.h, which, _p(out))); return out

    def sweep_gradhess_finish(self, want_cost=True):
        if not want_cost:
            self._chk(self.L.nlls_sweep_gradhess_finish(self.h, None)); return None
        return self._scalar(self.L.nlls_sweep_gradhess_finish)

    def solve_local(self):
        self._chk(self.L.nlls_solve_local(self.h))

    def solve_finish(self):
        self._chk(self.L.nlls_solve_finish(self.h, None))

    def reduce_buffer(self, stage):
        ptr, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.nlls_get_reduce_buffer(self.h, stage, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def time_sweep_gradhess(self, reps=10):
        ms = C.c_float(); self._chk(self.L.nlls_time_sweep_gradhess(self.h, reps, C.byref(ms))); return ms.value

    def time_sweep_accumulate(self, reps=10):
        ms = C.c_float(); self._chk(self.L.nlls_time_sweep_accumulate(self.h, reps, C.byref(ms))); return ms.value

    def time_sweep_cost(self, reps=10):
        ms = C.c_float(); self._chk(self.L.nlls_time_sweep_cost(self.h, reps, C.byref(ms))); return ms.value

    def profile_sweep(self, on=True, read=False):
        """In-situ timing of the accumulate launches (event pairs inside the caller's loop): read=True -> (avg, min, max ms, samples)."""
        if not read:
            self._chk(self.L.nlls_profile_sweep(self.h, 1 if on else 0, None, None, None, None)); return None
        a, mn, mx, n = C.c_float(), C.c_float(), C.c_float(), C.c_int64()
        self._chk(self.L.nlls_profile_sweep(self.h, 1 if on else 0, C.byref(a), C.byref(mn), C.byref(mx), C.byref(n)))
        return a.value, mn.value, mx.value, n.value

    def profile_sweep_dispatch(self):
        """(avg, min, max ms, samples) of the recorded accumulate launches by their dispatch timestamps (what a kernel trace reports); read before profile_sweep(False, read=True) resets nothing"""
        a, mn, mx, n = C.c_float(), C.c_float(), C.c_float(), C.c_int64()
        self._chk(self.L.nlls_profile_sweep_dispatch(self.h, C.byref(a), C.byref(mn), C.byref(mx), C.byref(n)))
        return a.value, mn.value, mx.value, n.value

    def time_reduced_solve(self, reps=3):
        ms = C.c_float(); self._chk(self.L.nlls_time_reduced_solve(self.h, reps, C.byref(ms))); return ms.value

    def time_solve(self, reps=3):
        ms = C.c_float(); self._chk(self.L.nlls_time_solve(self.h, reps, C.byref(ms))); return ms.value
