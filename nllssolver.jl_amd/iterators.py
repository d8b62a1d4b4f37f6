"""Host-side iterators: Newton, Levenberg-Marquardt, dogleg, gradient descent -- a line-for-line
behavioural mirror of src/iterators.jl; they stay on the host (BASELINE north_star) and drive the
device through the generic functions of MultiVariateLSgpu."""
import math
import sys
import time

import numpy as np

from ._capi import VARS_CURRENT, VARS_NEXT, NllsError, ERR_NOT_SPD

FLOATMIN = sys.float_info.min


def _timed(data, field, fn):
    t0 = time.perf_counter_ns()
    out = fn()
    setattr(data, field, getattr(data, field) + time.perf_counter_ns() - t0)
    return out


class NewtonData:                                   # src/iterators.jl:11-13
    def reset(self):
        return self

    def printable(self):
        return None


def iterate_newton(nd, data, problem, options):     # src/iterators.jl:15-27
    ls = data.linsystem
    _timed(data, "timesolver", ls.solve)
    data.linearsolvers += 1
    ls.update(VARS_NEXT, VARS_CURRENT)
    cost_ = _timed(data, "timecost", lambda: ls.cost(VARS_NEXT))
    data.costcomputations += 1
    return cost_


class LevMarData:                                   # src/iterators.jl:120-129
    def __init__(self):
        self.lambda_ = 0.0

    def reset(self):
        self.lambda_ = 0.0

    def printable(self):
        return 1.0 / self.lambda_ if self.lambda_ else math.inf


def iterate_levmar(lmd, data, problem, options):    # src/iterators.jl:139-172
    assert lmd.lambda_ >= 0.0
    ls = data.linsystem
    if lmd.lambda_ == 0:
        lmd.lambda_ = ls.initlambda()               # :142-144
    lastlambda = 0.0
    mu = 2.0
    while True:
        # A trial whose damped factorisation meets an exactly ZERO pivot (a singular direction at a tiny lambda): the reference's
        # LDLFactorizations would throw here; this path counts it as a rejected trial -- more damping, solve again -- and records it in
        # data.singulartrials (NLLSResult.singulartrials), in both branches.  No cost was computed for such a trial.
        singular = None
        if hasattr(ls, "lm_trial"):                   # :149-157 in one library call (same kernels, one synchronisation)
            try:
                cost_ = _timed(data, "timesolver", lambda: ls.lm_trial(lmd.lambda_ - lastlambda))
                data.costcomputations += 1
            except NllsError as e:
                if e.code != ERR_NOT_SPD:
                    raise
                singular = e
            lastlambda = lmd.lambda_
            data.linearsolvers += 1
        else:
            ls.uniformscaling(lmd.lambda_ - lastlambda)  # :149
            lastlambda = lmd.lambda_
            try:
                _timed(data, "timesolver", ls.solve)     # :152
            except NllsError as e:
                if e.code != ERR_NOT_SPD:
                    raise
                singular = e
            data.linearsolvers += 1
            if not singular:
                ls.update(VARS_NEXT, VARS_CURRENT)       # :155
                cost_ = _timed(data, "timecost", lambda: ls.cost(VARS_NEXT))   # :157
                data.costcomputations += 1
        if singular and (not math.isfinite(lmd.lambda_) or not np.all(np.isfinite(ls.b))):
            # ... unless the linear system ITSELF holds NaN / Inf (a non-finite residual): no damping cures that.  The reference's factorisation
            # does not throw on it -- its step and the trial cost come out NaN, '!(cost_ > bestcost)' accepts them and the outer loop reports
            # "cost is NaN" + "NaN in the step" (src/optimize.jl:147-152).  Same here: a NaN step, retracted, its cost swept.
            ls.x = np.full(len(ls.b), np.nan)
            ls.update(VARS_NEXT, VARS_CURRENT)
            cost_ = _timed(data, "timecost", lambda: ls.cost(VARS_NEXT))
            data.costcomputations += 1
            singular = None
        if singular:
            if not math.isfinite(lmd.lambda_ * mu):      # nothing left to damp with
                raise singular
            data.singulartrials += 1
            lmd.lambda_ *= mu
            mu *= 2.0
            continue
        if not (cost_ > data.bestcost) or ls.step_maxabs() < options.dstep:   # :160
            ls.uniformscaling(-lastlambda)           # :162
            xHx, gx = ls.quadform()
            stepquality = (cost_ - data.bestcost) / (0.5 * xHx + gx)          # :163
            lmd.lambda_ *= (1 - (2 * stepquality - 1) ** 3) if stepquality < 0.983 else 0.1   # :164
            return cost_
        lmd.lambda_ *= mu                            # :169-170
        mu *= 2.0


class DoglegData:                                   # src/iterators.jl:30-45
    def __init__(self):
        self.trustradius = 0.0
        self.cauchy = None

    def reset(self):
        self.trustradius = 0.0

    def printable(self):
        return self.trustradius


def iterate_dogleg(dd, data, problem, options):     # src/iterators.jl:47-115
    ls = data.linsystem
    t0 = time.perf_counter_ns()
    gradient = ls.b
    gnorm2 = float(gradient @ gradient)
    a = gnorm2 / (ls.grad_quadform() + FLOATMIN)
    dd.cauchy = -a * gradient
    alpha2 = a * a * gnorm2
    alpha = math.sqrt(alpha2)
    if dd.trustradius == 0:
        dd.trustradius = alpha
    beta = 0.0
    x = None
    if alpha < dd.trustradius:
        ls.solve()
        x = np.array(ls.x)
        beta = float(np.linalg.norm(x))
        data.linearsolvers += 1
    data.timesolver += time.perf_counter_ns() - t0
    cost_ = data.bestcost
    while True:
        if not (alpha < dd.trustradius):
            x = (dd.trustradius / alpha) * dd.cauchy
            linear_approx = dd.trustradius * (2 * alpha - dd.trustradius) / (2 * a)
        elif beta <= dd.trustradius:
            linear_approx = cost_
        else:
            x = x - dd.cauchy
            sq_leg = float(x @ x)
            c = float(dd.cauchy @ x)
            trsq = dd.trustradius * dd.trustradius - alpha2
            step = math.sqrt(c * c + sq_leg * trsq)
            step = (-c + step) / sq_leg if c <= 0 else trsq / (c + step)
            x = x * step + dd.cauchy
            linear_approx = 0.5 * (a * (1 - step) ** 2 * gnorm2) + step * (2 - step) * cost_
        ls.x = x
        ls.update(VARS_NEXT, VARS_CURRENT)
        cost_ = _timed(data, "timecost", lambda: ls.cost(VARS_NEXT))
        data.costcomputations += 1
        mu = (data.bestcost - cost_) / linear_approx
        if mu > 0.375:
            dd.trustradius = max(dd.trustradius, 3 * float(np.linalg.norm(x)))
        elif mu < 0.125:
            dd.trustradius *= 0.5
        if not (cost_ > data.bestcost) or float(np.max(np.abs(x))) < options.dstep:
            return cost_


class GradientDescentData:                          # src/iterators.jl:177-185
    def __init__(self):
        self.stepsize = 1.0

    def reset(self):
        self.stepsize = 1.0

    def printable(self):
        return self.stepsize


def iterate_gradientdescent(gd, data, problem, options):   # src/iterators.jl:187-208
    ls = data.linsystem
    gradient = ls.b
    x = -gradient * gd.stepsize
    ls.x = x
    ls.update(VARS_NEXT, VARS_CURRENT)
    costc = _timed(data, "timecost", lambda: ls.cost(VARS_NEXT))
    data.costcomputations += 1
    while costc > data.bestcost:
        coststep = float(x @ gradient)
        costdiff = data.bestcost + coststep - costc
        gd.stepsize *= 0.5 * coststep / costdiff
        x = -gradient * gd.stepsize
        ls.x = x
        ls.update(VARS_NEXT, VARS_CURRENT)
        costc = _timed(data, "timecost", lambda: ls.cost(VARS_NEXT))
        data.costcomputations += 1
    gd.stepsize *= 2
    return costc
