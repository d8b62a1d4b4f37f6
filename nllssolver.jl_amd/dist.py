"""Observation sharding across the GPUs of one node, one process per GPU.

Residual blocks are partitioned BY ELIMINATED VARIABLE (bundle adjustment: by point): a rank owns a
contiguous range of points and every cost block that touches them, so the point block-rows of A.data
(99.8 % of its bytes) are written by exactly one rank and never communicated.  What is summed over ranks
(torch.distributed: RCCL over xGMI on the GPU box, gloo in the CPU tests):
  stage 0  after the gradient sweep : [cost | reduced-block rows of A.data | reduced part of b]   (~0.3 MB)
  stage 1  after local elimination  : [S (dense, lower) | s]  -- the real collective of this path
  stage 2  after back-substitution  : x  (each rank contributes its own points)                    (~2.4 MB)
plus one scalar per cost sweep; dogleg / gradient descent also sum the gradient b once per iteration (~2.4 MB).
"""
import numpy as np

from . import _capi
from .linearsystem import MultiVariateLSgpu


class _DevArray:
    """Wraps a raw device pointer for torch.as_tensor via __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class ShardedLS(MultiVariateLSgpu):
    """MultiVariateLSgpu whose sweeps/solves are sharded over `world` ranks (one process per GPU).

    dist: torch.distributed (initialised by the caller: "nccl" = RCCL on the GPU box).  host_staged=True routes the
    buffer reductions through host copies (gloo), which lets several ranks share ONE GPU in tests."""

    def __init__(self, problem, unfixed, flags=0, device=0, rank=0, world=1, dist=None, host_staged=False):
        self.rank, self.world, self.dist, self.host_staged = rank, world, dist, host_staged
        self._pre_upload = (rank, world)
        super().__init__(problem, unfixed, flags, device)
        sh = self.ctx.shard_info()
        self.local_nobs, self.local_nnz_data, self.local_ndof_written = sh["local_ncost"], sh["local_nnz_data"], sh["local_ndof"]

    def _make_context(self, device):
        ctx = _capi.Context(device)
        rank, world = self._pre_upload
        if world > 1:
            ctx.set_shard(rank, world)
        return ctx

    # ---- collectives ------------------------------------------------------------------------------
    def _allreduce_buffer(self, stage):
        import torch
        ptr, n = self.ctx.reduce_buffer(stage)
        if n == 0:
            return
        t = torch.as_tensor(_DevArray(ptr, n), device="cuda")
        if self.host_staged:       # gloo on host copies: lets two ranks share one GPU in tests
            h = t.cpu(); self.dist.all_reduce(h); t.copy_(h)
        else:
            self.dist.all_reduce(t)
        torch.cuda.synchronize()

    def _allreduce_scalars(self, values, op="sum"):
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if self.host_staged else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(v) for v in t.cpu()]

    def costgradhess(self, want_cost=True):
        if self.world == 1:
            return super().costgradhess(want_cost)
        self._x = None
        self.ctx.sweep_gradhess_local()
        self._allreduce_buffer(0)
        return self.ctx.sweep_gradhess_finish()

    def cost(self, which=_capi.VARS_NEXT):
        c = super().cost(which)                    # this rank's cost blocks only
        return c if self.world == 1 else self._allreduce_scalars([c])[0]

    def lm_trial(self, dlambda):
        if self.world == 1:
            return super().lm_trial(dlambda)
        self.uniformscaling(dlambda); self.solve(); self.update(_capi.VARS_NEXT, _capi.VARS_CURRENT)   # sharded: the separate steps
        return self.cost(_capi.VARS_NEXT)

    def solve(self):
        if self.world == 1:
            return super().solve()
        self._x = None
        self.ctx.solve_local()
        self._allreduce_buffer(1)
        self.ctx.solve_finish()
        self._allreduce_buffer(2)

    def initlambda(self):
        m = self.ctx.max_abs_diag()
        if self.world > 1:
            m = self._allreduce_scalars([m], "max")[0]
        return m * 1e-6

    def quadform(self):
        a, g = self.ctx.quadform()
        return (a, g) if self.world == 1 else tuple(self._allreduce_scalars([a, g]))

    def grad_quadform(self):
        g = self.ctx.grad_quadform()               # this rank's rows of H against the (complete) gradient
        return g if self.world == 1 else self._allreduce_scalars([g])[0]

    @property
    def b(self):
        """The full gradient on every rank (dogleg, gradient descent: src/iterators.jl:48,191): each rank contributes the
        rows it owns (nlls_get_grad_owned), one sum over ranks."""
        if self.world == 1:
            return self.ctx.get_grad()
        import torch
        t = torch.from_numpy(self.ctx.get_grad_owned())
        if not self.host_staged:
            t = t.cuda()
        self.dist.all_reduce(t)
        return t.cpu().numpy()


def partition_by_weight(weights, nparts):
    """Contiguous ranges [start_k, start_{k+1}) over len(weights) items with balanced weight sums
    (the same rule csrc/nlls_structure.cpp uses to split the eliminated blocks over ranks)."""
    w = np.asarray(weights, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for k in range(1, nparts):
        target = (total * k) // nparts
        bounds.append(int(np.searchsorted(cum, target, side="left")))
    bounds.append(len(w))
    return np.maximum.accumulate(np.array(bounds, dtype=np.int64))
