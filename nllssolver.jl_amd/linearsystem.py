"""MultiVariateLSgpu: the device-resident linear system that replaces MultiVariateLSsparse /
MultiVariateLSdense (src/linearsystem.jl:44-87) behind the same generic functions the iterators call
(SURVEY.md 8b).  All arithmetic happens in csrc/libnlls_amd.so; this class only forwards.
"""
import numpy as np

from . import _capi
from ._capi import VARS_CURRENT, VARS_NEXT, VARS_BEST


class MultiVariateLSgpu:
    def __init__(self, problem, unfixed, flags=0, device=0, stream=None):
        unfixed = np.asarray(unfixed, dtype=bool)
        assert unfixed.size == problem.nvariables
        # blockindices: linearsystem.jl:93-102
        self.blockindices = np.zeros(problem.nvariables, np.uint64)
        self.blockindices[unfixed] = np.arange(1, int(unfixed.sum()) + 1, dtype=np.uint64)
        self.ctx = self._make_context(device)
        if stream is not None:
            self.ctx.set_stream(stream)
        self.info = self.ctx.upload(problem.var_kind, problem.var_dim, self.blockindices, problem.groups(), flags)
        self.ctx.set_variables(problem.variables, VARS_CURRENT)
        self.ctx.copy_variables(VARS_NEXT, VARS_CURRENT)        # setupiterator: varnext = deepcopy(variables)
        self._x = None

    def _make_context(self, device):
        return _capi.Context(device)

    # ---- the generic functions of SURVEY 8b -------------------------------------------------------
    def costgradhess(self, want_cost=True):
        """zero!(linsystem); costgradhess!(linsystem, vars, costs)   src/optimize.jl:118,167-170.
        want_cost=False (the outer loop between iterations discards the value): enqueue only."""
        self._x = None
        return self.ctx.sweep_gradhess(want_cost)

    def cost(self, which=VARS_NEXT):
        """cost(vars, costs)   src/cost.jl:10-13"""
        return self.ctx.sweep_cost(which)

    def uniformscaling(self, k):
        """uniformscaling!(hessian, k)   src/iterators.jl:149,162"""
        self.ctx.damp(k)

    def solve(self):
        """negate!(solve!(linsystem, options))   src/iterators.jl:152"""
        self._x = None
        self.ctx.solve()

    def lm_trial(self, dlambda):
        """uniformscaling!(H, dlambda); solve!; update!(next, current); cost(next)   src/iterators.jl:149-157, fused
        into one call of the library (one synchronisation).  Returns the trial cost."""
        self._x = None
        return self.ctx.lm_trial(dlambda, VARS_NEXT, VARS_CURRENT)

    def update(self, to=VARS_NEXT, frm=VARS_CURRENT):
        """update!(to, from, linsystem)   src/linearsystem.jl:206-213"""
        self.ctx.retract(to, frm)

    def initlambda(self):
        """src/iterators.jl:131-137"""
        return self.ctx.max_abs_diag() * 1e-6

    def quadform(self):
        """(fast_bAb(hessian, x), dot(gradient, x))   src/iterators.jl:163"""
        return self.ctx.quadform()

    def grad_quadform(self):
        """fast_bAb(hessian, gradient)   src/iterators.jl:52"""
        return self.ctx.grad_quadform()

    def step_maxabs(self):
        return self.ctx.step_maxabs()

    def step_norm(self):
        return self.ctx.step_norm()

    @property
    def x(self):
        """linsystem.x, fetched lazily (callbacks read it: src/callbacks.jl:47,105)"""
        if self._x is None:
            self._x = self.ctx.get_step()
        return self._x

    @x.setter
    def x(self, value):
        self._x = np.array(value, dtype=np.float64)
        self.ctx.set_step(self._x)

    @property
    def b(self):
        return self.ctx.get_grad()

    def swap(self, a, b):
        self.ctx.swap_variables(a, b)

    def copy(self, dst, src):
        self.ctx.copy_variables(dst, src)

    def variables(self, which=VARS_CURRENT):
        return self.ctx.get_variables(which)

    def close(self):
        self.ctx.close()


def makesymmvls(problem, unfixed, flags=0, device=0, stream=None):
    """makesymmvls(problem, unfixed, nblocks)   src/linearsystem.jl:91-124"""
    return MultiVariateLSgpu(problem, unfixed, flags, device, stream)
