"""Observation sharding across the GPUs of one node, one process per GPU.

Residual blocks are partitioned BY ELIMINATED VARIABLE (bundle adjustment: by point): a rank owns a
contiguous range of points and every cost block that touches them, so the point block-rows of A.data
(99.8 % of its bytes) are written by exactly one rank and never communicated.  What is summed over ranks
(torch.distributed: RCCL over xGMI on the GPU box, gloo in the CPU tests):
  stage 0  after the gradient sweep : [cost | reduced-block rows of A.data | reduced part of b]   (~0.3 MB, once per iteration)
  stage 1  after local elimination  : [S | s] in band layout -- the real collective of this path   (3.4 MB at config 4, once per LM trial)
A Levenberg-Marquardt trial needs nothing else but ONE gather of the trial's scalars per rank, on the device (cost, x'Hx, g'x, max|x|, |x|^2 and the
factorisation status -- every rank raises on the reduced status, so a rank-local bad pivot cannot leave the ranks in different
collectives): the reduced system is factorised on every rank, so every rank holds the reduced part of the step, retracts the
reduced variables and its own eliminated ones itself and sweeps its own cost blocks -- the step x is never summed (stage 2, 2.4 MB,
is only taken by plain solve() calls: Newton, dogleg).  Dogleg / gradient descent also sum the gradient b once per iteration.
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

    def __init__(self, problem, unfixed, flags=0, device=0, rank=0, world=1, dist=None, host_staged=False, force_collectives=False, presharded=False):
        """presharded: `problem` is already THIS rank's share (all reduced variables + its own eliminated ones, its own cost blocks:
        synthetic.create_ba_problem_shard) -- NLLS_FLAG_PRESHARDED; nothing is partitioned by the library."""
        self.rank, self.world, self.dist, self.host_staged = rank, world, dist, host_staged
        self.device = int(device)
        if presharded and world > 1:
            flags |= _capi.FLAG_PRESHARDED
        # force_collectives: take the local / reduce / finish route even with one rank (rehearses the RCCL plumbing on one GPU)
        self.sharded = world > 1 or (force_collectives and dist is not None)
        self._trial = None                         # ((x'Hx, g'x), max|x|, |x|) of the last lm_trial, until the step changes
        self._views, self._gathered = {}, None
        self._pre_upload = (rank, world)
        super().__init__(problem, unfixed, flags, device)
        self._tstream = None
        self.device = int(device)
        assert not (world > 1 and dist is None), "ShardedLS with world > 1 needs an initialised torch.distributed"
        if self.sharded and not host_staged:
            # RCCL mode: the library works on a torch stream, and the collectives are issued under that stream -- kernels and
            # all-reduces are then ordered by the stream itself and no device-wide synchronisation is needed between them
            import torch
            self._tstream = torch.cuda.Stream(device=self.device)
            self.ctx.set_stream(self._tstream.cuda_stream)
        # The collectives of the LM loop live BEHIND the C ABI (include/nlls_amd.h): RCCL inside the library on the GPU box, or -- host_staged -- a
        # callback that sums host copies with gloo so that several ranks can share one GPU in the tests.  costgradhess / initlambda /
        # lm_trial / cost below are then plain calls of the same entry points as on one GPU, and the library's own outer loop
        # (nlls_lm_iterations) drives the sharded trials: no Python between two trials.  The Python route further down (solve_local /
        # all-reduce / solve_finish) remains for plain solve() calls (Newton, dogleg: they want x on every rank).
        self.native_collectives = self.sharded                 # (installed in _make_context: a pre-sharded upload is itself collective)
        sh = self.ctx.shard_info()
        # REPLICAS (include/nlls_amd.h, nlls_get_shard_info): the problem does not shard -- every rank runs the whole of it and holds the complete result; nothing below may
        # enter a collective (agree_max still does: every rank calls it)
        self.replicated = bool(sh.get("replicated", 0))
        if self.replicated:
            self.sharded = False; self.native_collectives = False
        self.local_nobs, self.local_nnz_data, self.local_ndof_written = sh["local_ncost"], sh["local_nnz_data"], sh["local_ndof"]

    def _make_context(self, device):
        ctx = _capi.Context(device)
        rank, world = self._pre_upload
        if world > 1:
            ctx.set_shard(rank, world)
        if self.sharded:
            if self.host_staged:
                ctx.set_allreduce(self._staged_allreduce)
            else:
                ids = [_capi.Context.comm_unique_id() if rank == 0 else None]
                if world > 1:
                    self.dist.broadcast_object_list(ids, src=0)
                ctx.comm_init_rccl(ids[0])
        return ctx

    def _staged_allreduce(self, ptr, count, op, stream):
        """nlls_allreduce_fn for the tests: gloo on host copies (synchronises the stream on both sides)"""
        import torch
        torch.cuda.synchronize()
        t = torch.as_tensor(_DevArray(ptr, count), device=f"cuda:{self.device}")
        h = t.cpu()
        if self.world > 1:
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM)
        t.copy_(h)
        torch.cuda.synchronize()
        return 0

    # ---- collectives ------------------------------------------------------------------------------
    def _buffer_tensor(self, stage):
        """torch view of one of the library's reduce buffers (device memory; wrapped once per buffer)"""
        import torch
        ptr, n = self.ctx.reduce_buffer(stage)
        key = (stage, ptr, n)
        if key not in self._views:
            self._views[key] = torch.as_tensor(_DevArray(ptr, n), device=f"cuda:{self.device}") if n > 0 else None
        return self._views[key]

    def _allreduce_buffer(self, stage):
        import torch
        ptr, n = self.ctx.reduce_buffer(stage)
        if n == 0:
            return
        if self.host_staged:       # gloo on host copies: lets two ranks share one GPU in tests
            torch.cuda.synchronize()                   # the local phase was only enqueued (on the library's own stream)
            t = torch.as_tensor(_DevArray(ptr, n), device=f"cuda:{self.device}")
            h = t.cpu(); self.dist.all_reduce(h); t.copy_(h)
            torch.cuda.synchronize()
        else:
            with torch.cuda.stream(self._tstream):
                self.dist.all_reduce(self._buffer_tensor(stage))

    def _allreduce_scalars(self, values, op="sum"):
        import torch
        op = self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM
        if self.host_staged:
            t = torch.tensor(list(values), dtype=torch.float64); self.dist.all_reduce(t, op=op)
        else:
            with torch.cuda.stream(self._tstream):
                t = torch.tensor(list(values), dtype=torch.float64).cuda(self.device, non_blocking=True); self.dist.all_reduce(t, op=op); t = t.cpu()
        return [float(v) for v in t]

    def costgradhess(self, want_cost=True):
        if not self.sharded:
            return super().costgradhess(want_cost)
        if self.native_collectives:
            self._x = None; self._trial = None
            return MultiVariateLSgpu.costgradhess(self, want_cost)          # collective inside the library
        self._x = None; self._trial = None
        self.ctx.sweep_gradhess_local()
        self._allreduce_buffer(0)
        return self.ctx.sweep_gradhess_finish(want_cost)

    def cost(self, which=_capi.VARS_NEXT):
        c = super().cost(which)                    # this rank's cost blocks only -- summed inside the library under native collectives
        return c if (not self.sharded or self.native_collectives) else self._allreduce_scalars([c])[0]

    def lm_trial(self, dlambda):
        if not self.sharded:
            return super().lm_trial(dlambda)
        if self.native_collectives:
            self._x = None; self._trial = None
            return MultiVariateLSgpu.lm_trial(self, dlambda)                # nlls_lm_trial: the whole sharded trial in one call (csrc/nlls_capi.cpp)
        # sharded: damp, local elimination, ONE buffer reduction ([S | s]), replicated reduced solve + own back-substitution, then
        # retraction + cost sweep + step statistics in one call and ONE gather of six scalars per rank; what the iterator asks next
        # (quadform, step_maxabs) is answered from here.  No reduction of x: every rank holds the reduced part of the step.
        self.uniformscaling(dlambda)
        self._x = None; self._trial = None
        self.ctx.solve_local()
        self._allreduce_buffer(1)
        self.ctx.solve_finish_replicated()                 # enqueue only: the status comes home with the trial's scalars
        if self.host_staged:
            out = self.ctx.trial_local(_capi.VARS_NEXT, _capi.VARS_CURRENT)
            allv = self._allgather_scalars(out)            # [world][6]
        else:
            # RCCL: the trial is only enqueued, its scalars are gathered on the device -- ONE synchronisation per trial (the copy home)
            import torch
            self.ctx.trial_local_enqueue(_capi.VARS_NEXT, _capi.VARS_CURRENT)
            src = self._buffer_tensor(3)
            with torch.cuda.stream(self._tstream):
                if self._gathered is None:
                    self._gathered = torch.empty((self.world, src.numel()), dtype=torch.float64, device=f"cuda:{self.device}")
                self.dist.all_gather_into_tensor(self._gathered, src)
                raw = self._gathered.cpu().numpy()
            allv = raw[:, [0, 8, 5, 1, 9 if self.world > 1 else 2, 10]]     # cost, x'Hx, g'x, max|x|, |x|^2 (own share), status
        status = int(allv[:, 5].max())
        if status != 0:                                    # raised on EVERY rank, from the reduced value
            raise _capi.NllsError(_capi.ERR_NOT_SPD, f"factorisation met a zero pivot on some rank (code {status})")
        c, a, g = allv[:, 0].sum(), allv[:, 1].sum(), allv[:, 2].sum()
        self._trial = ((float(a), float(g)), float(allv[:, 3].max()), float(np.sqrt(allv[:, 4].sum())))
        return float(c)

    def _allgather_scalars(self, values):
        import torch
        v = torch.tensor(np.asarray(values, dtype=np.float64))
        if self.host_staged:
            outl = [torch.zeros_like(v) for _ in range(self.world)]
            if self.world > 1: self.dist.all_gather(outl, v)
            else: outl = [v]
            return torch.stack(outl).numpy()
        with torch.cuda.stream(self._tstream):
            vd = v.cuda(self.device, non_blocking=True)
            outd = torch.empty((self.world, v.numel()), dtype=torch.float64, device=f"cuda:{self.device}")
            self.dist.all_gather_into_tensor(outd, vd)
            return outd.cpu().numpy()

    def variables(self, which=_capi.VARS_CURRENT):
        """The complete variable set: every rank contributes the variables it has kept up to date (its own eliminated blocks'; rank 0
        also the reduced ones), one sum over ranks."""
        if not self.sharded or self.world == 1:
            return super().variables(which)
        import torch
        t = torch.from_numpy(self.ctx.get_variables_owned(which))
        if self.host_staged:
            self.dist.all_reduce(t)
            return t.numpy()
        with torch.cuda.stream(self._tstream):
            t = t.cuda(self.device); self.dist.all_reduce(t); t = t.cpu()
        return t.numpy()

    def solve(self, _in_trial=False):
        if not self.sharded:
            return super().solve()
        self._x = None; self._trial = None
        self.ctx.solve_local()
        self._allreduce_buffer(1)
        # the factorisation status is made collective before anyone raises: a zero pivot in a rank's own eliminated blocks is seen
        # by that rank only, and it must still take part in the stage-2 reduction its peers enter
        err = None
        try:
            self.ctx.solve_finish()
        except _capi.NllsError as e:
            if e.code != _capi.ERR_NOT_SPD:
                raise
            err = e
        self._allreduce_buffer(2)
        if self._allreduce_scalars([1.0 if err else 0.0], "max")[0] != 0:
            raise err if err else _capi.NllsError(_capi.ERR_NOT_SPD, "factorisation met a zero pivot on another rank")

    def agree_max(self, value):
        """the maximum over ranks of a rank-local scalar (the Python outer loop's deadline flag: src/optimize.jl:158 under sharding)"""
        return self._allreduce_scalars([float(value)], "max")[0] if self.world > 1 else float(value)

    def initlambda(self):
        m = self.ctx.max_abs_diag()
        if self.sharded and not self.native_collectives:
            m = self._allreduce_scalars([m], "max")[0]
        return m * 1e-6

    @property
    def x(self):
        return MultiVariateLSgpu.x.fget(self)

    @x.setter
    def x(self, value):
        self._trial = None
        MultiVariateLSgpu.x.fset(self, value)

    def step_maxabs(self):
        return self._trial[1] if self._trial else super().step_maxabs()

    def step_norm(self):
        return self._trial[2] if self._trial else super().step_norm()

    def quadform(self):
        if self._trial:
            return self._trial[0]
        a, g = self.ctx.quadform()
        return (a, g) if (not self.sharded or self.native_collectives) else tuple(self._allreduce_scalars([a, g]))

    def grad_quadform(self):
        g = self.ctx.grad_quadform()               # this rank's rows of H against the (complete) gradient
        return g if (not self.sharded or self.native_collectives) else self._allreduce_scalars([g])[0]

    @property
    def b(self):
        """The full gradient on every rank (dogleg, gradient descent: src/iterators.jl:48,191): each rank contributes the
        rows it owns (nlls_get_grad_owned), one sum over ranks."""
        if not self.sharded:
            return self.ctx.get_grad()
        import torch
        t = torch.from_numpy(self.ctx.get_grad_owned())
        if self.host_staged:
            self.dist.all_reduce(t)
            return t.numpy()
        with torch.cuda.stream(self._tstream):
            t = t.cuda(self.device); self.dist.all_reduce(t); t = t.cpu()
        return t.numpy()


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
