"""Worker of tests/test_gpu_sharded.py: rank `r` of `w` on ONE GPU, gloo + host-staged reductions.
Checks the sharded sweep / solve / LM against the unsharded device path on the same problem."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"     # "nccl": RCCL plumbing with ONE rank (device buffers, no host staging)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        assert world == 1
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=__import__('datetime').timedelta(seconds=90))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__('datetime').timedelta(seconds=90))
    staged = backend != "nccl"
    # one scenario per seed; a comma-separated list runs them one after the other in ONE process group (the start of a rank -- interpreter, torch, rendezvous -- is most of a
    # scenario's two seconds): every scenario ends with a barrier and `destroy_process_group`, which is held back until the last one
    seeds = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "21").split(",")]
    real_destroy = dist.destroy_process_group
    if len(seeds) > 1:
        dist.destroy_process_group = lambda: None
    for seed in seeds:
        scenario(rank, world, dist, staged, seed)
        sys.stdout.flush()
    if len(seeds) > 1:
        dist.barrier(); real_destroy()


def scenario(rank, world, dist, staged, seed):
    mk = lambda: ShardedLS(p, unfixed, rank=rank, world=world, dist=dist, host_staged=staged, force_collectives=True)
    import nllssolver_jl_amd as N
    from nllssolver_jl_amd import synthetic, _capi, iterators as It, optimizer as Opt
    from nllssolver_jl_amd.dist import ShardedLS
    from nllssolver_jl_amd.linearsystem import MultiVariateLSgpu

    if seed < 0:
        return singular_point_block(rank, world, dist, staged)
    if 5000 <= seed < 6000:
        return presharded(rank, world, dist, staged, seed)
    if 7000 <= seed < 8000:
        return deadline(rank, world, dist, staged)
    if 10000 <= seed < 11000:
        return replicas(rank, world, dist, staged, seed)
    grid = 8000 <= seed < 9000                   # a 24 x 24 camera grid with permuted labels: the reduced system goes to the tile-sparse solver (nested dissection at upload: every rank must arrive at the same tiles)
    shuffled = 6000 <= seed < 8000               # the cameras' labels permuted: the reduced system is re-ordered at upload (every rank must arrive at the same order)
    rng = np.random.default_rng(seed)
    ncr = int(rng.integers(12, 90))             # (sharding partitions by eliminated block: stay within what the Schur kernels take)
    shape = (40, 2000, 0.15) if seed == 21 else (90, 3000, 0.08) if shuffled else (ncr, int(rng.integers(300, 4000)), float(rng.uniform(4.0, 10.0)) / ncr)
    config4 = seed == 4000                       # BASELINE config 4 at full size (1k cameras x 100k points x 1M residual blocks: the configuration north_star shards), against the ORACLE
    config3 = seed == 3000 or config4            # BASELINE config 3 at full size, checked against the ORACLE (not only the unsharded device)
    if config3:
        shape = (1000, 100000, 0.01) if config4 else (100, 10000, 0.1)
    shuf = (lambda q: synthetic.shuffle_camera_labels(q, shape[0], seed)) if shuffled else (lambda q: q)
    mkp = lambda: synthetic.perturb_ba_problem(shuf(synthetic.create_ba_problem(*shape, seed=1 if config3 else seed, robust=N.HuberKernel(0.01 if config3 else 0.05),
                                                                                outlier_frac=0.05, outlier_sigma=0.05)), 1e-3, 1e-3)
    if 9000 <= seed < 10000:
        # a bundle adjustment with three dynamic-size variables beside it (src/autodiff.jl:96-121): their cost blocks hold no eliminated variable -- owned round robin --
        # and their diagonal blocks are reduced rows, summed over ranks like the cameras'
        from nllssolver_jl_amd import kinds as K
        def mkp():
            q = synthetic.perturb_ba_problem(synthetic.create_ba_problem(40, 2000, 0.15, seed=seed, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
            r = np.random.default_rng(seed)
            for n in (20, 70, 33):
                v = q.addvariable(0.3 * r.standard_normal(n), K.VAR_DYNAMIC); X = r.standard_normal(n); X /= np.linalg.norm(X)
                q.addcosts(K.RES_DYN_LINEAR, [[v]], np.concatenate([[1.0], X])[None, :]); q.addcosts(K.RES_DYN_NORM, [[v]], np.zeros((1, 0)))
            return q
    if grid:
        mkp = lambda: synthetic.perturb_ba_problem(synthetic.shuffle_camera_labels(synthetic.create_grid_ba_problem(24, 24, 4, seed=3, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05,
                                                                                                                    noise=1e-3), 576, seed), 1e-3, 1e-3)
    p = mkp()
    unfixed = np.ones(p.nvariables, bool)
    ref = MultiVariateLSgpu(p, unfixed)                       # unsharded reference on the same device
    sh = mk()
    if grid:
        st_r, st_s = ref.ctx.solve_stats(), sh.ctx.solve_stats()
        assert ref.info.solve_mode == 3 and sh.info.solve_mode == 3 and all(st_r[k] == st_s[k] for k in ("tsp_tiles", "tsp_levels", "tsp_lower_tiles")), (st_r, st_s)
        mi = sh.ctx.memory_info()       # only the tiles the assembly writes into (and s) are summed over ranks: the fill tiles are zero on every rank until the factorisation
        assert 0 < mi["sharded_reduce_bytes"] < mi["reduced_system_bytes"], mi
    if shuffled:
        st_r, st_s = ref.ctx.solve_stats(), sh.ctx.solve_stats()
        assert st_r["reordered"] == 1 and st_s["reordered"] == 1 and ref.info.solve_mode == 2 and sh.info.solve_mode == 2 and sh.info.bandwidth == ref.info.bandwidth, (st_r, st_s)
    info = sh.ctx.shard_info()
    counts = [None] * world
    dist.all_gather_object(counts, info["local_ncost"])
    assert sum(counts) == p.ncosts(), (counts, p.ncosts())      # every cost block owned exactly once
    assert max(counts) - min(counts) <= 0.1 * p.ncosts() / world + 64, counts

    c_ref, c_sh = ref.costgradhess(), sh.costgradhess()
    assert np.isclose(c_ref, c_sh, rtol=1e-12), (c_ref, c_sh)
    assert np.isclose(ref.cost(_capi.VARS_CURRENT), sh.cost(_capi.VARS_CURRENT), rtol=1e-12)
    lam = ref.initlambda(); assert np.isclose(lam, sh.initlambda(), rtol=1e-13)
    ref.uniformscaling(lam); sh.uniformscaling(lam)
    ref.solve(); sh.solve()
    x_ref, x_sh = ref.x, sh.x
    assert np.max(np.abs(x_ref - x_sh)) < 1e-8 * np.max(np.abs(x_ref)), np.max(np.abs(x_ref - x_sh))
    q_ref, q_sh = ref.quadform(), sh.quadform()
    assert np.allclose(q_ref, q_sh, rtol=1e-9), (q_ref, q_sh)
    assert np.isclose(ref.step_maxabs(), sh.step_maxabs(), rtol=1e-9)

    if config3:
        # the same sweep, damped solve and four LM iterations in the CPU oracle (every rank checks its own view of the result)
        from tests.helpers import oracle_problem, blockindices
        op = oracle_problem(mkp()); ols = op.linear_system(blockindices(p))
        c_or = ols.costgradhess()
        assert np.isclose(c_or, c_sh, rtol=1e-11), (c_or, c_sh)
        b_or = ols.b.copy()
        assert np.isclose(ols.max_abs_diag() * 1e-6, lam, rtol=1e-12)
        ols.solve(lam); x_or = ols.x
        assert np.max(np.abs(x_or - x_sh)) < 1e-7 * np.max(np.abs(x_or)), np.max(np.abs(x_or - x_sh))
        assert np.max(np.abs(b_or - sh.b)) < 1e-10 * np.max(np.abs(b_or))

    if seed == 21:
        # optimizesingles! under sharding (round 5; src/optimize.jl:60-76): every rank passes the same lists and relaxes the points it owns -- all their cost blocks are
        # local --, one gather of the results: bit for bit what the unsharded device does to the same points; cameras (their blocks spread over ranks) are declined everywhere
        pts = np.nonzero(p.var_dim == 3)[0] + 1
        cptr, cgroup, cindex, cslot = p.costlists(pts)
        o = N.NLLSOptions()
        kw = dict(maxiters=o.maxiters, maxfails=o.maxfails, reldcost=o.reldcost, absdcost=o.absdcost, dstep=o.dstep, iterator=int(o.iterator))
        for c_ in (ref.ctx, sh.ctx): c_.set_variables(p.variables, _capi.VARS_CURRENT)
        it_r = ref.ctx.optimize_singles(pts, cptr, cgroup, cindex, cslot, **kw)
        it_s = sh.ctx.optimize_singles(pts, cptr, cgroup, cindex, cslot, **kw)
        v_r, v_s = ref.ctx.get_variables(_capi.VARS_CURRENT), sh.ctx.get_variables(_capi.VARS_CURRENT)
        assert np.array_equal(it_r, it_s) and it_r.min() >= 1, (it_r[:8], it_s[:8])
        assert np.array_equal(v_r, v_s) and not np.array_equal(v_r, p.variables), np.max(np.abs(v_r - v_s))
        if world > 1:
            cams = np.nonzero(p.var_dim == 6)[0] + 1              # (every camera: those at the ranks' boundaries see points of two ranks)
            try:
                sh.ctx.optimize_singles(cams, *p.costlists(cams), **kw); raise AssertionError("cameras relaxed under sharding")
            except _capi.NllsError as e:
                assert e.code == _capi.ERR_UNSUPPORTED, e
        for c_ in (ref.ctx, sh.ctx): c_.set_variables(p.variables, _capi.VARS_CURRENT)
        ref.costgradhess(); sh.costgradhess()
    # the full gradient is assembled from the ranks' own rows (dogleg / gradient descent need it on every rank)
    b_ref, b_sh = ref.b, sh.b
    assert np.max(np.abs(b_ref - b_sh)) < 1e-10 * np.max(np.abs(b_ref)), np.max(np.abs(b_ref - b_sh))
    assert np.isclose(ref.grad_quadform(), sh.grad_quadform(), rtol=1e-9)

    # a few full iterations through the host loop on both: Levenberg-Marquardt, then dogleg
    def run(ls, iters=4, itdata=It.LevMarData, itfn=It.iterate_levmar, native=False):
        opts = N.NLLSOptions(maxiters=iters)
        data = Opt.NLLSInternal(ls, time.perf_counter_ns())
        loop = Opt.OuterLoop(p, opts, data, itdata(), itfn, N.nullcallback, native=native)
        loop.start()
        if native:          # nlls_lm_iterations: the library's own outer loop drives the SHARDED trials (collectives behind the C ABI)
            assert loop.native, "the sharded linear system must qualify for the library's outer loop"
            while loop.iterations(1 << 30) == 0:
                pass
        else:
            while loop.iteration() == 0:
                pass
        assert data.iternum > 1, data.iternum                  # (a loop that stops after one iteration compares nothing)
        return data.bestcost, ls.variables(_capi.VARS_CURRENT)
    ref2 = MultiVariateLSgpu(p, unfixed); sh2 = mk()
    cr, vr = run(ref2); cs, vs = run(sh2)
    assert np.isclose(cr, cs, rtol=1e-9), (cr, cs)
    sh4 = mk(); cs4, vs4 = run(sh4, native=True)
    # lazy stage 0: inside the library's loop the reduced rows are summed over ranks ONCE (the initial damping asks for max |diag|); every later trial runs on
    # the ranks' shares (nlls_get_solve_stats [11], [12])
    st4 = sh4.ctx.solve_stats(); sh4.close()
    assert world == 1 or (st4["lazy_trials"] >= 1 and st4["reduced_row_sums"] <= 2), st4
    assert np.isclose(cs4, cs, rtol=1e-9), (cs4, cs)           # ... and the same loop inside the library, every rank in it
    assert np.allclose(vs4, vs, rtol=1e-6, atol=1e-9)
    if config3:
        ores = oracle_problem(mkp()).optimize(maxiters=4)
        assert np.isclose(ores.bestcost, cs, rtol=1e-8), (ores.bestcost, cs)
    if config4:     # (no dogleg leg at this size: the single steps, the LM loops -- host and library -- and the oracle's loop are what this scenario is for)
        for o in (ref, sh, ref2, sh2):
            o.close()
        dist.barrier(); dist.destroy_process_group()
        print(f"rank {rank}: sharded == unsharded (BASELINE config 4 at full size against the oracle, cost {cs:.6e}, owned {info['local_ncost']} of {p.ncosts()} cost blocks)")
        return
    if grid:        # (no dogleg leg: its undamped steps on this gauge-free problem need the pivot floor only the band solver has)
        for o in (ref, sh, ref2, sh2):
            o.close()
        dist.barrier(); dist.destroy_process_group()
        print(f"rank {rank}: sharded == unsharded (tile-sparse reduced solver, cost {cs:.6e}, owned {info['local_ncost']} of {p.ncosts()} cost blocks)")
        return
    ref3 = MultiVariateLSgpu(p, unfixed); sh3 = mk()
    cd_r, _ = run(ref3, 4, It.DoglegData, It.iterate_dogleg); cd_s, _ = run(sh3, 4, It.DoglegData, It.iterate_dogleg)
    # (dogleg takes UNDAMPED Newton steps: on the gauge-free affine camera H is singular and the step along the gauge directions is decided by
    # rounding -- the summation order of the elimination's atomics -- so two runs of the SAME unsharded loop differ at the 1e-3 level after a
    # few iterations.  What is comparable: both reduce the cost, to the same level.  Exact agreement of single steps is asserted above.)
    # The tolerance is what two UNSHARDED runs show on this very problem (their spread, measured here), not a constant.
    ref5 = MultiVariateLSgpu(p, unfixed); cd_r2, _ = run(ref5, 4, It.DoglegData, It.iterate_dogleg); ref5.close()
    # (every rank has run the unsharded loop twice in its own process: the spread over ALL of these runs -- two runs of one process agree more closely
    #  than runs of different processes do)
    runs = [None] * world; dist.all_gather_object(runs, (cd_r, cd_r2)); runs = [v for pair in runs for v in pair]
    spread = (max(runs) - min(runs)) / abs(cd_r)
    assert cd_r < 0.9 * c_ref and cd_s < 0.9 * c_ref and abs(cd_r - cd_s) <= max(8.0 * spread, 1e-6) * abs(cd_r), (c_ref, cd_r, cd_r2, cd_s, spread)
    for o in (ref, sh, ref2, sh2, ref3, sh3):
        o.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: sharded == unsharded (cost {cs:.6e}, owned {info['local_ncost']} of {p.ncosts()} cost blocks)")


def presharded(rank, world, dist, staged, seed):
    """NLLS_FLAG_PRESHARDED: every rank uploads ONLY its share (all cameras + its own points and their cost blocks, its own variable numbering);
    the library's outer loop over the collective entry points must reach what the unsharded device reaches on the whole problem."""
    import nllssolver_jl_amd as N
    from nllssolver_jl_amd import synthetic, _capi, iterators as It, optimizer as Opt
    from nllssolver_jl_amd.dist import ShardedLS
    from nllssolver_jl_amd.linearsystem import MultiVariateLSgpu
    ncam, npts = 60, 3000
    shuffled = seed % 1000 >= 100               # camera labels permuted: the ranks agree on the reverse Cuthill-McKee order of the UNION of their camera graphs
    grid = seed % 1000 >= 300                   # a 24 x 24 camera grid (labels permuted): the tile-sparse solver's nested dissection runs on the UNION of the ranks' camera graphs
    if grid:
        ncam, npts = 576, 576 * 4
        p = synthetic.create_grid_ba_problem(24, 24, 4, seed=3, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05, noise=1e-3)
    else:
        p = synthetic.create_ba_problem(ncam, npts, 0.12, seed=seed, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05)
    if shuffled:
        p = synthetic.shuffle_camera_labels(p, ncam, seed)
    p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
    mine = synthetic.shard_of_problem(p, ncam, rank, world)
    if 200 <= seed % 1000 < 300:
        # the share with its POINTS listed before the cameras: the cameras' rows of A.data then hold the E blocks, as many as this rank has observations --
        # the ranks' stage-0 buffers would differ in length and be summed element by element.  The upload must refuse it, on every rank alike.
        g = next(iter(mine.costs.values())); vi, da = g.arrays(); nl = mine.nvariables - ncam
        q = N.NLLSProblem(); q.addvariables(mine.variables[6 * ncam:].reshape(nl, 3)); q.addvariables(mine.variables[: 6 * ncam].reshape(ncam, 6))
        q.addcosts(g.res_kind, np.stack([vi[:, 0] + nl, vi[:, 1] - ncam], axis=1), da, g.robust)
        try:
            ShardedLS(q, np.ones(q.nvariables, bool), rank=rank, world=world, dist=dist, host_staged=staged, presharded=True)
            raise AssertionError("a pre-sharded upload whose reduced rows differ between ranks was accepted")
        except _capi.NllsError as e:
            assert e.code == _capi.ERR_INVALID_ARG and "reduced rows differ" in str(e), str(e)
        dist.barrier(); dist.destroy_process_group()
        print(f"rank {rank}: sharded == unsharded (mismatched pre-sharded layout refused on every rank)")
        return
    counts = [None] * world
    dist.all_gather_object(counts, mine.ncosts()); assert sum(counts) == p.ncosts(), (counts, p.ncosts())
    def run(ls, prob, native):
        data = Opt.NLLSInternal(ls, time.perf_counter_ns())
        loop = Opt.OuterLoop(prob, N.NLLSOptions(maxiters=5), data, It.LevMarData(), It.iterate_levmar, N.nullcallback, native=native)
        loop.start()
        while (loop.iterations(1 << 30) if native else loop.iteration()) == 0:
            pass
        return data
    ref = MultiVariateLSgpu(p, np.ones(p.nvariables, bool)); dr = run(ref, p, True)
    sh = ShardedLS(mine, np.ones(mine.nvariables, bool), rank=rank, world=world, dist=dist, host_staged=staged, presharded=True)
    assert sh.info.nreduced_dof == ref.info.nreduced_dof and sh.info.bandwidth == ref.info.bandwidth, (sh.info.bandwidth, ref.info.bandwidth)   # the agreed layout
    if grid:
        st_r, st_s = ref.ctx.solve_stats(), sh.ctx.solve_stats()
        assert ref.info.solve_mode == 3 and sh.info.solve_mode == 3 and all(st_r[k] == st_s[k] for k in ("tsp_tiles", "tsp_levels", "tsp_lower_tiles")), (st_r, st_s)
    elif shuffled:
        assert sh.ctx.solve_stats()["reordered"] == 1 and sh.info.solve_mode == 2, (sh.ctx.solve_stats(), sh.info.solve_mode)
    ds = run(sh, mine, True)
    assert np.isclose(ds.startcost, dr.startcost, rtol=1e-12) and np.isclose(ds.bestcost, dr.bestcost, rtol=1e-9), (ds.bestcost, dr.bestcost)
    assert ds.iternum == dr.iternum and ds.linearsolvers == dr.linearsolvers
    vr, vs = ref.variables(_capi.VARS_CURRENT), sh.ctx.get_variables(_capi.VARS_CURRENT)
    assert np.allclose(vs[: 6 * ncam], vr[: 6 * ncam], rtol=1e-7, atol=1e-10)                       # every rank holds the cameras
    l0, l1 = (npts * rank) // world, (npts * (rank + 1)) // world
    assert np.allclose(vs[6 * ncam:], vr[6 * ncam + 3 * l0: 6 * ncam + 3 * l1], rtol=1e-7, atol=1e-10)   # ... and its own points
    # the generator of a rank's share alone: the same structure as the slice of the whole (counts per rank), nothing else built
    if not shuffled:
        g = synthetic.create_ba_problem_shard(ncam, npts, 0.12, rank, world, seed=seed)
        assert g.ncosts() == mine.ncosts() and g.nvariables == mine.nvariables
        assert np.array_equal(next(iter(g.costs.values())).arrays()[0], next(iter(mine.costs.values())).arrays()[0])
    ref.close(); sh.close(); dist.barrier(); dist.destroy_process_group()
    print(f"rank {rank}: sharded == unsharded (presharded: {mine.ncosts()} of {p.ncosts()} cost blocks uploaded, best cost {ds.bestcost:.6e})")


def replicas(rank, world, dist, staged, seed):
    """Problems that do not shard (round 5): a DENSE system (a curve fit: 3000 scalar residuals over four scalar variables; a three-camera bundle adjustment whose whole system is
    dense) under nlls_set_shard(rank, world > 1) used to be refused ("sharding needs the block-sparse path").  Now: replicas -- every rank runs the whole problem as rank 0 of 1,
    no entry point enters a collective (an installed all-reduce is never called), nlls_get_shard_info()[5] = world; every rank reaches the unsharded result bit for bit
    in its own sums' order (rtol 1e-12), through the library's outer loop and the Python one, and holds the complete variable set."""
    import nllssolver_jl_amd as N
    from nllssolver_jl_amd import synthetic, _capi, iterators as It, optimizer as Opt
    from nllssolver_jl_amd.dist import ShardedLS
    from nllssolver_jl_amd.linearsystem import MultiVariateLSgpu
    for which in ("curvefit", "tiny_ba"):
        mkp = (lambda: synthetic.create_curvefit_problem(3000, seed=seed)[0]) if which == "curvefit" else \
              (lambda: synthetic.perturb_ba_problem(synthetic.create_ba_problem(3, 8, 1.0, seed=seed), 1e-3, 1e-3))
        for native in (True, False):
            p, q = mkp(), mkp()
            unfixed = np.ones(p.nvariables, bool)
            sh = ShardedLS(q, unfixed, rank=rank, world=world, dist=dist, host_staged=staged, force_collectives=True)
            info = sh.ctx.shard_info()
            assert sh.replicated == (world > 1) and info["replicated"] == (world if world > 1 else 0) and info["nranks"] == 1 and not sh.info.is_sparse, info
            ref = MultiVariateLSgpu(p, unfixed)
            out = []
            for ls, prob in ((ref, p), (sh, q)):
                opts = N.NLLSOptions(maxiters=12)
                data = Opt.NLLSInternal(ls, time.perf_counter_ns())
                loop = Opt.OuterLoop(prob, opts, data, It.LevMarData(), It.iterate_levmar, N.nullcallback, native=native)
                loop.start()
                while (loop.iterations(1 << 30) if loop.native else loop.iteration()) == 0:
                    pass
                assert data.iternum > 1
                out.append((data.bestcost, data.iternum, ls.variables(_capi.VARS_CURRENT).copy()))
            assert out[0][1] == out[1][1] and np.isclose(out[0][0], out[1][0], rtol=1e-12) and np.allclose(out[0][2], out[1][2], rtol=1e-9, atol=1e-12), (out[0][:2], out[1][:2])
            assert out[1][0] < 0.5 * N.cost(mkp())
            ref.close(); sh.close()
    dist.barrier(); dist.destroy_process_group()
    print(f"rank {rank}: sharded == unsharded (replicas: dense systems run whole on every one of {world} ranks, no collective)")


def deadline(rank, world, dist, staged):
    """src/optimize.jl:158 under sharding: `time > starttime + maxtime` is rank-local (every process has its own clock and its own start time).  Rank 0 is
    given a deadline it crosses after a few iterations, the other ranks one they never reach: all ranks must leave the loop in the SAME iteration with
    the time-out bit set -- a rank that left alone would leave its peers inside the next trial's collectives (a hang).  Both loops: the library's
    (nlls_lm_iterations: the flags ride in the trial's scalar gather) and the Python one (one all-reduce of the flag)."""
    import nllssolver_jl_amd as N
    from nllssolver_jl_amd import synthetic, _capi, iterators as It, optimizer as Opt
    from nllssolver_jl_amd.dist import ShardedLS
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(40, 2000, 0.15, seed=21, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
    for native in (True, False):
        sh = ShardedLS(p, np.ones(p.nvariables, bool), rank=rank, world=world, dist=dist, host_staged=staged, force_collectives=True)
        opts = N.NLLSOptions(maxiters=10 ** 6, reldcost=-np.inf, absdcost=-np.inf, dstep=-1.0, maxfails=10 ** 9, maxtime=0.15 if rank == 0 else 1e6)
        dist.barrier()
        data = Opt.NLLSInternal(sh, time.perf_counter_ns())
        loop = Opt.OuterLoop(p, opts, data, It.LevMarData(), It.iterate_levmar, N.nullcallback, native=native)
        loop.start()
        assert loop.native == native
        conv = 0
        while conv == 0:
            conv = loop.iterations(1 << 30) if native else loop.iteration()
        seen = [None] * world; dist.all_gather_object(seen, (int(data.iternum), int(conv)))
        assert all(v == seen[0] for v in seen), seen              # the same iteration, the same flags, on every rank
        assert conv == 1 << 9 and data.iternum >= 2, (conv, data.iternum)
        sh.close()
    dist.barrier(); dist.destroy_process_group()
    print(f"rank {rank}: sharded == unsharded (deadline agreed: all ranks left in iteration {seen[0][0]})")


def singular_point_block(rank, world, dist, staged):
    """A zero pivot that only ONE rank sees (an eliminated block whose diagonal block is exactly zero): every rank must raise
    ERR_NOT_SPD -- from the reduced status -- instead of one rank leaving the collectives its peers are in (a hang)."""
    import nllssolver_jl_amd as N
    from nllssolver_jl_amd import synthetic, _capi
    from nllssolver_jl_amd.dist import ShardedLS
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(30, 1200, 0.2, seed=5), 1e-3, 1e-3)
    # the LAST point is seen by the first camera only, and that camera's pose is all zeros: the point's Jacobian -- and so its
    # diagonal block C_v -- is exactly zero.  The last point belongs to the last rank.
    g = next(iter(p.costs.values())); vi, da = g.arrays()
    last = p.nvariables
    keep = vi[:, 1] != last
    vi2 = np.concatenate([vi[keep], [[1, last]]]); da2 = np.concatenate([da[keep], [[0.0, 0.0]]])
    g.set_arrays(vi2, da2)
    p.variables[:6] = 0.0
    sh = ShardedLS(p, np.ones(p.nvariables, bool), rank=rank, world=world, dist=dist, host_staged=staged, force_collectives=True)
    sh.costgradhess()
    raised = 0
    try:
        sh.solve()                                  # undamped: the zero block meets the factorisation as it is
    except _capi.NllsError as e:
        assert e.code == _capi.ERR_NOT_SPD; raised += 1
    try:
        sh.lm_trial(0.0)
    except _capi.NllsError as e:
        assert e.code == _capi.ERR_NOT_SPD; raised += 1
    assert raised == 2, raised                      # on EVERY rank, although only one rank owns the singular block
    c = sh.lm_trial(1e-3 * sh.initlambda() * 1e6)   # and the ranks are still in step: a damped trial goes through
    assert np.isfinite(c)
    sh.close(); dist.barrier(); dist.destroy_process_group()
    print(f"rank {rank}: singular block raised on every rank")


if __name__ == "__main__":
    main()
