"""Worker of tests/test_dist_cpu.py: world_size-2 gloo on CPU.  Checks the sharding rule (costs follow the rank that
owns their eliminated block) with the CPU oracle standing in for the kernels: per-rank partial costs / gradients of
the reduced blocks must sum (all-reduce) to the unsharded ones."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__('datetime').timedelta(seconds=90))
    import nllssolver_jl_amd as N
    from nllssolver_jl_amd import synthetic, kinds as K
    from nllssolver_jl_amd.dist import partition_by_weight
    from oracle import oracle as O

    ncam, npts = 12, 300
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, 0.3, seed=5), 1e-3, 1e-3)
    (grp,) = p.costs.values(); vi, da = grp.arrays()
    # weights: cost blocks per point; contiguous ranges of points of (nearly) equal weight
    w = np.bincount(vi[:, 1] - ncam - 1, minlength=npts)
    bounds = partition_by_weight(w, world)
    assert bounds[0] == 0 and bounds[-1] == npts and np.all(np.diff(bounds) >= 0)
    owner = np.searchsorted(bounds, vi[:, 1] - ncam - 1, side="right") - 1
    mine = owner == rank
    counts = [None] * world; dist.all_gather_object(counts, int(mine.sum()))
    assert sum(counts) == vi.shape[0] and max(counts) - min(counts) <= w.max() + 1
    # the full problem and this rank's share of the cost blocks, same variables
    full = O.OracleProblem(p.var_kind, p.var_dim, p.groups()); full.set_variables(p.variables)
    q = N.NLLSProblem(); q.addvariables(p.variables[:6 * ncam].reshape(ncam, 6)); q.addvariables(p.variables[6 * ncam:].reshape(npts, 3))
    q.addcosts(K.RES_BA_AFFINE, vi[mine], da[mine])
    part = O.OracleProblem(q.var_kind, q.var_dim, q.groups()); part.set_variables(q.variables)
    # cost: partial sums add up
    t = torch.tensor([part.cost()], dtype=torch.float64); dist.all_reduce(t)
    assert np.isclose(t.item(), full.cost(), rtol=1e-13)
    # gradient / diagonal blocks of the cameras (the reduced blocks): summed over ranks == unsharded; points: owner only
    bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    lf = full.linear_system(bi, 4); lp = part.linear_system(bi, 4)      # FORCE_SPARSE: same layout family
    lf.costgradhess(); lp.costgradhess()
    bc = torch.tensor(lp.b[:6 * ncam].copy()); dist.all_reduce(bc)
    assert np.allclose(bc.numpy(), lf.b[:6 * ncam], rtol=1e-12, atol=1e-14)
    own_pts = np.unique(vi[mine][:, 1]) - ncam - 1
    for j in own_pts[:50]:
        assert np.allclose(lp.b[6 * ncam + 3 * j: 6 * ncam + 3 * j + 3], lf.b[6 * ncam + 3 * j: 6 * ncam + 3 * j + 3], rtol=1e-12, atol=1e-14)
    dist.barrier(); dist.destroy_process_group()
    print(f"rank {rank}: ok ({int(mine.sum())} of {vi.shape[0]} cost blocks)")


if __name__ == "__main__":
    main()
