"""optimize! and its result/option structs: a behavioural mirror of src/optimize.jl:5-17,78-180 and
src/structs.jl:5-107 on the host, driving the device-resident linear system."""
import enum
import math
import time

import numpy as np

from . import iterators as It
from ._capi import VARS_BEST, VARS_CURRENT, VARS_NEXT
from .callbacks import nullcallback
from .linearsystem import makesymmvls


class NLLSIterator(enum.IntEnum):                   # src/structs.jl:5
    newton = 0
    levenbergmarquardt = 1
    dogleg = 2
    gradientdescent = 3


newton, levenbergmarquardt, dogleg, gradientdescent = NLLSIterator


class NLLSOptions:                                   # src/structs.jl:22-35
    def __init__(self, maxiters=100, reldcost=1e-15, absdcost=1e-15, dstep=1e-15, maxfails=3, maxtime=30.0,
                 iterator=levenbergmarquardt, callback=None, iteratordata=None):
        self.reldcost, self.absdcost, self.dstep = reldcost, absdcost, dstep
        self.maxfails, self.maxiters = maxfails, maxiters
        self.maxtime = int(round(maxtime * 1e9))     # nanoseconds
        self.iterator, self.callback, self.iteratordata = NLLSIterator(iterator), callback, iteratordata


class NLLSResult:                                    # src/structs.jl:37-79
    def __init__(self, d):
        self.startcost, self.bestcost = d.startcost, d.bestcost
        self.timetotal, self.timeinit = d.timetotal * 1e-9, d.timeinit * 1e-9
        self.timecost, self.timegradient, self.timesolver = d.timecost * 1e-9, d.timegradient * 1e-9, d.timesolver * 1e-9
        self.termination, self.niterations = d.converged, d.iternum
        self.costcomputations, self.gradientcomputations, self.linearsolvers = d.costcomputations, d.gradientcomputations, d.linearsolvers
        self.singulartrials = d.singulartrials      # LM trials rejected because the damped factorisation met an exactly zero pivot (the reference would throw)

    def __str__(self):
        other = self.timetotal - self.timecost - self.timegradient - self.timesolver - self.timeinit
        tt = max(self.timetotal, 1e-300)
        s = ("NLLSsolver optimization took %f seconds and %d iterations to reduce the cost from %e to %e (a %.2f%% reduction), using:\n"
             "   %d cost computations in %f seconds (%.2f%% of total time),\n"
             "   %d gradient computations in %f seconds (%.2f%% of total time),\n"
             "   %d linear solver computations in %f seconds (%.2f%% of total time),\n"
             "   %f seconds for initialization (%.2f%% of total time), and\n"
             "   %f seconds for other stuff (%.2f%% of total time).\n") % (
            self.timetotal, self.niterations, self.startcost, self.bestcost, 100 * (1 - self.bestcost / self.startcost) if self.startcost else 0.0,
            self.costcomputations, self.timecost, 100 * self.timecost / tt, self.gradientcomputations, self.timegradient, 100 * self.timegradient / tt,
            self.linearsolvers, self.timesolver, 100 * self.timesolver / tt, self.timeinit, 100 * self.timeinit / tt, other, 100 * other / tt)
        reasons = ["Cost is infinite.", "Cost is NaN.", "Relative decrease in cost below threshold.", "Absolute decrease in cost below threshold.",
                   "Step contains an infinite value.", "Step contains a NaN.", "Step size below threshold.",
                   "Too many consecutive iterations increasing the cost.", "Maximum number of outer iterations reached.",
                   "Maximum allowed computation time exceeded."]
        if self.termination:
            s += "Reason(s) for termination:\n"
            for bit, r in enumerate(reasons):
                if self.termination & (1 << bit):
                    s += "   " + r + "\n"
            if self.termination >> 16:
                s += "   Terminated by user-defined callback, with flags: " + bin(self.termination >> 16)[2:] + "\n"
        return s


class NLLSInternal:                                  # src/structs.jl:81-104
    def __init__(self, linsystem, starttime):
        self.startcost = self.bestcost = 0.0
        self.starttime = starttime
        self.timetotal = self.timeinit = self.timecost = self.timegradient = self.timesolver = 0
        self.iternum = self.costcomputations = self.gradientcomputations = self.linearsolvers = self.converged = 0
        self.singulartrials = 0
        self.linsystem = linsystem


_ITER = {
    newton: (It.NewtonData, It.iterate_newton),
    levenbergmarquardt: (It.LevMarData, It.iterate_levmar),
    dogleg: (It.DoglegData, It.iterate_dogleg),
    gradientdescent: (It.GradientDescentData, It.iterate_gradientdescent),
}


def convertunfixed(unfixed, problem):                # src/optimize.jl:19-22
    n = problem.nvariables
    if unfixed is None:
        return np.ones(n, bool)
    if isinstance(unfixed, (int, np.integer)) and not isinstance(unfixed, (bool, np.bool_)):
        m = np.zeros(n, bool); m[int(unfixed) - 1] = True   # 1-based single variable index
        return m
    if isinstance(unfixed, tuple) and len(unfixed) == 2 and not isinstance(unfixed[0], (bool, np.bool_)):
        kind, dim = unfixed                          # "variable type": (kind, dim)
        return (problem.var_kind == kind) & (problem.var_dim == dim)
    return np.asarray(unfixed, dtype=bool)


class OuterLoop:
    """State of optimizeinternal! (src/optimize.jl:109-180), split so that bench.py can time exactly K outer
    iterations: start() = :111-121, iteration() = one pass of the while-loop body :124-171."""

    def __init__(self, problem, options, data, iteratedata, iterate, callback, native=None):
        self.problem, self.options, self.data = problem, options, data
        self.iteratedata, self.iterate, self.callback = iteratedata, iterate, callback
        self.fails = 0
        self.have_best = False
        self.cost = math.nan
        # native: Levenberg-Marquardt without a per-iteration callback on one GPU runs in the library's own host loop
        # (nlls_lm_iterations, csrc/nlls_lm.cpp: the same statements as iteration() + iterate_levmar below, without the interpreter
        # between two trials).  native=False keeps the Python loop (the tests hold the two against each other).
        ls = data.linsystem
        can = (iterate is It.iterate_levmar and callback is nullcallback and hasattr(ls, "ctx")
               and (not getattr(ls, "sharded", False) or getattr(ls, "native_collectives", False))     # (sharded: the collectives are inside the library)
               and hasattr(ls.ctx, "lm_iterations"))
        self.native = can if native is None else (bool(native) and can)
        self._state = None

    def start(self):
        data, ls = self.data, self.data.linsystem
        data.startcost = -math.inf                       # preoptimization, src/iterators.jl:7
        self.fails = 0
        data.iternum = 0
        self.stoptime = data.starttime + self.options.maxtime
        data.timeinit += time.perf_counter_ns() - data.starttime
        cost = It._timed(data, "timegradient", ls.costgradhess)      # :118
        data.gradientcomputations += 1
        data.bestcost = cost
        data.startcost = max(cost, data.startcost)
        self.cost = cost

    def iterations(self, n):
        """n outer iterations (or fewer, if one of them terminates); returns the termination flags of the last one."""
        if not self.native:
            conv = 0
            for _ in range(n):
                conv = self.iteration()
                if conv:
                    break
            return conv
        from . import _capi
        data, ls, o = self.data, self.data.linsystem, self.options
        big = 2 ** 62
        clamp = lambda v: int(max(-big, min(big, v)))
        opt = _capi.LmOptions(float(o.reldcost), float(o.absdcost), float(o.dstep), clamp(o.maxfails), clamp(o.maxiters), clamp(self.stoptime))
        if self._state is None:
            self._state = _capi.LmState()
        st = self._state
        st.lambda_, st.bestcost, st.cost = float(self.iteratedata.lambda_), float(data.bestcost), float(self.cost)
        st.iternum, st.fails, st.have_best = int(data.iternum), int(self.fails), int(self.have_best)
        st.linearsolvers, st.costcomputations, st.gradientcomputations, st.singulartrials = data.linearsolvers, data.costcomputations, data.gradientcomputations, data.singulartrials
        st.timesolver_ns = st.timegradient_ns = st.timecost_ns = 0
        ls._x = None
        try:
            ls.ctx.lm_iterations(opt, st, n)
        finally:
            self.iteratedata.lambda_ = st.lambda_
            data.bestcost, self.cost, data.iternum, self.fails, self.have_best = st.bestcost, st.cost, st.iternum, st.fails, bool(st.have_best)
            data.linearsolvers, data.costcomputations, data.gradientcomputations, data.singulartrials = st.linearsolvers, st.costcomputations, st.gradientcomputations, st.singulartrials
            data.timesolver += st.timesolver_ns; data.timegradient += st.timegradient_ns; data.timecost += st.timecost_ns      # (device times where the trials' launches time themselves: nlls_get_time_buckets)
            data.converged = st.converged
        return int(st.converged)

    def iteration(self, regrad=True):
        """One outer iteration; returns the termination flags (0 = keep going).  With regrad the linear
        problem for the next iteration is built unless terminating (:167-170)."""
        if self.native and regrad:
            return self.iterations(1)
        data, ls, options = self.data, self.data.linsystem, self.options
        data.iternum += 1
        cost = float(self.iterate(self.iteratedata, data, self.problem, options))   # :126
        if self.callback is not nullcallback:
            # a user callback sees the trial point where the reference puts it -- problem.varnext (src/optimize.jl:128) -- and what it writes there counts: the
            # reference's EM callback rewrites the kernel variable in varnext (test/adaptivecost.jl:15-25, src/robustadaptive.jl:48-73) and updatefromnext! then
            # makes it the current point.  So: the next variables up before the callback, and down again after it when they changed.
            self.problem.varnext = ls.variables(VARS_NEXT)
            before = self.problem.varnext.copy()
        cost, terminate = self.callback(cost, self.problem, data, self.iteratedata)   # :128
        if self.callback is not nullcallback and not np.array_equal(before, self.problem.varnext, equal_nan=True):
            ls.ctx.set_variables(np.ascontiguousarray(self.problem.varnext, np.float64), VARS_NEXT)
        dcost = data.bestcost - cost
        if dcost >= 0:
            data.bestcost = cost
            self.fails = 0
        else:
            dcost = cost
            self.fails += 1
            if self.fails == 1:                          # :137-144 store the current best variables
                if self.have_best:
                    ls.swap(VARS_CURRENT, VARS_BEST)
                else:
                    ls.copy(VARS_BEST, VARS_CURRENT); self.have_best = True
        ls.swap(VARS_CURRENT, VARS_NEXT)                 # updatefromnext!  :207-209
        maxstep = ls.step_maxabs()                       # :149
        converged = 0
        converged |= int(math.isinf(cost)) << 0
        converged |= int(math.isnan(cost)) << 1
        converged |= int(dcost < data.bestcost * options.reldcost) << 2
        converged |= int(dcost < options.absdcost) << 3
        converged |= int(math.isinf(maxstep)) << 4
        converged |= int(math.isnan(maxstep)) << 5
        converged |= int(maxstep < options.dstep) << 6
        converged |= int(self.fails > options.maxfails) << 7
        converged |= int(data.iternum >= options.maxiters) << 8
        late = time.perf_counter_ns() > self.stoptime
        if getattr(ls, "sharded", False) and hasattr(ls, "agree_max"):
            late = ls.agree_max(float(late)) > 0          # every rank has its own clock: all leave in the same iteration (one all-reduce of a flag)
        converged |= int(late) << 9
        converged |= int(terminate) << 16
        data.converged = converged
        self.cost = cost
        if converged == 0 and regrad:
            It._timed(data, "timegradient", lambda: ls.costgradhess(want_cost=False))   # :167-170 (the value is discarded there too)
            data.gradientcomputations += 1
        return converged

    def finish(self):
        data, ls = self.data, self.data.linsystem
        if not (data.bestcost >= self.cost):
            ls.swap(VARS_CURRENT, VARS_BEST)             # updatefrombest!  :173-176
        data.timetotal += time.perf_counter_ns() - data.starttime
        return data


def optimizeinternal(problem, options, data, iteratedata, iterate, callback, native=None):   # src/optimize.jl:109-180
    loop = OuterLoop(problem, options, data, iteratedata, iterate, callback, native)
    loop.start()
    while loop.iterations(1 << 30 if loop.native else 1) == 0:
        pass
    return loop.finish()


def optimize(problem, options=None, unfixed=None, callback=nullcallback, flags=0, device=0, stream=None, native=None):
    """optimize!(problem, options, unfixed, callback) -> NLLSResult   src/optimize.jl:5-17,57.
    Variables are optimised in place: problem.variables holds the best values on return."""
    options = options or NLLSOptions()
    starttime = time.perf_counter_ns()
    assert problem.nvariables > 0
    unfixed = convertunfixed(unfixed, problem)
    ls = makesymmvls(problem, unfixed, flags, device, stream)       # :16
    data = NLLSInternal(ls, starttime)
    mk, iterate = _ITER[options.iterator]
    try:
        optimizeinternal(problem, options, data, mk(), iterate, callback or nullcallback, native)
        problem.variables[:] = ls.variables(VARS_CURRENT)
    finally:
        ls.close()
    return NLLSResult(data)


def optimizesingles(problem, options=None, indices=None, kind=None, dim=None, device=0):
    """optimizesingles!(problem, options, indices | type)   src/optimize.jl:60-76,183-205: every listed variable is
    optimised on its own (all others fixed) against the cost blocks that depend on it.  `indices` 1-based, or select by
    variable kind (and dimension), like the reference's `type` argument.  Any of the four iterators; one GPU thread per variable.
    Listed variables that share a cost block are relaxed one after the other, as the reference does: in launches of independent sets
    (NLLSProblem.singles_levels).  Returns the iterations each variable took, in the order of `indices`."""
    options = options or NLLSOptions()
    if indices is None:
        sel = problem.var_kind == kind
        if dim is not None:
            sel &= problem.var_dim == dim
        indices = np.nonzero(sel)[0] + 1
    indices = np.asarray(indices, dtype=np.int64)
    # "sorted in order of variable size" (src/optimize.jl:67; sortperm is stable): the processing order
    from . import kinds as K
    vkind, vdim = problem.var_kind, problem.var_dim            # (properties that build an array per access: taken once)
    dof = np.array([K.var_dof(vkind[i - 1], vdim[i - 1]) for i in indices], dtype=np.int64) if indices.size else np.zeros(0, np.int64)
    order = np.argsort(dof, kind="stable")
    ordered = indices[order]
    iters = np.zeros(indices.size, np.int64)
    # What the one-thread-per-variable kernel takes: at most 6 degrees of freedom, fixed-size blocks, not the adaptive kernel's variable.  Anything else -- a
    # DynamicVector of run-time length, the kernel variable -- is relaxed the way the reference relaxes EVERY listed variable (src/optimize.jl:183-205): the
    # sub-problem of the cost blocks that depend on it, only that variable free, through the ordinary device path (a univariate dense system).  The listed
    # order is kept: runs of kernel-sized variables go in launches of independent sets, a wide variable in between is a sub-problem of its own.
    gl = list(problem.costs.values())
    adaptive_first = [gi for gi, g in enumerate(gl) if g.res_kind in K.ADAPTIVE_KINDS]
    # (the kernel variables of the adaptive groups, gathered ONCE: a scan of every adaptive group per listed variable was O(listed x blocks) -- 37 s of host time on ba_so3_500x50k)
    kernelvars = np.unique(np.concatenate([gl[gi].arrays()[0][:, 0] for gi in adaptive_first])) if adaptive_first else np.zeros(0, np.int64)
    if ordered.size:
        is_wide = (dof[order] > 6) | (vkind[ordered - 1] == K.VAR_DYNAMIC) | np.isin(ordered, kernelvars)
    else:
        is_wide = np.zeros(0, bool)
    ls = makesymmvls(problem, np.ones(problem.nvariables, bool), 0, device)
    try:
        q = 0
        while q < ordered.size:
            if is_wide[q]:
                v = int(ordered[q])
                cptr, cgroup, cindex, _ = problem.costlists(np.array([v]), check=False)
                from .problem import NLLSProblem
                sub = NLLSProblem(); sub.copy_variables_from(problem, ls.variables(VARS_CURRENT))
                for gi in sorted(set(cgroup.tolist())):
                    vi, da = gl[gi].arrays(); sel = cindex[cgroup == gi]
                    sub.addcosts(gl[gi].res_kind, vi[sel], da[sel], gl[gi].robust)
                unfixed = np.zeros(problem.nvariables, bool); unfixed[v - 1] = True
                res = optimize(sub, options, unfixed, device=device)
                ls.ctx.set_variables(sub.variables, VARS_CURRENT)
                iters[order[q]] = res.niterations
                q += 1
                continue
            r = q
            while r < ordered.size and not is_wide[r]:
                r += 1
            run = ordered[q:r]; level = problem.singles_levels(run)
            for lv in range(int(level.max()) + 1 if level.size else 0):
                pick = np.nonzero(level == lv)[0]
                cptr, cgroup, cindex, cslot = problem.costlists(run[pick])
                it = ls.ctx.optimize_singles(run[pick], cptr, cgroup, cindex, cslot, options.maxiters, options.maxfails,
                                             options.reldcost, options.absdcost, options.dstep, iterator=int(options.iterator))
                iters[order[q + pick]] = it
            q = r
        problem.variables[:] = ls.variables(VARS_CURRENT)
    finally:
        ls.close()
    return iters


def cost(problem, device=0):
    """cost(problem)   src/cost.jl:9"""
    ls = makesymmvls(problem, np.ones(problem.nvariables, bool), 0, device)
    try:
        return ls.cost(VARS_CURRENT)
    finally:
        ls.close()
