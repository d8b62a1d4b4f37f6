#!/usr/bin/env python3
"""K Levenberg-Marquardt iterations of a bench workload through the library's own loop (nlls_lm_iterations) and nothing else behind the upload: the target of the
rocprofv3 --pmc passes that count the HBM bytes of ONE LM iteration (tools/pmc_iter.sh), matrix-free (default) or materialised (--materialise)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi, iterators as It, optimizer as Opt
from nllssolver_jl_amd.dist import ShardedLS

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ba_1kx100k"); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--materialise", action="store_true")
a = ap.parse_args()
ncam, npts, prop = {"ba_100x10k": (100, 10000, 0.1), "ba_1kx100k": (1000, 100000, 0.01)}[a.workload]
p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
ls = ShardedLS(p, np.ones(p.nvariables, bool), flags=0, device=0, rank=0, world=1, dist=None, host_staged=False)
if a.materialise:
    ls.ctx.set_option(_capi.OPT_MATERIALIZE, 1)
options = N.NLLSOptions(maxiters=10 ** 9, reldcost=-np.inf, absdcost=-np.inf, dstep=-1.0, maxfails=10 ** 9, maxtime=1e6)
ls.ctx.set_variables(p.variables, _capi.VARS_CURRENT); ls.ctx.copy_variables(_capi.VARS_NEXT, _capi.VARS_CURRENT)
data = Opt.NLLSInternal(ls, time.perf_counter_ns())
loop = Opt.OuterLoop(p, options, data, It.LevMarData(), It.iterate_levmar, N.nullcallback); loop.start()
t0 = time.perf_counter(); loop.iterations(a.iters); dt = time.perf_counter() - t0
st = ls.ctx.solve_stats()
print(json.dumps({"workload": a.workload, "iterations": int(loop.data.iternum), "linear_solves": int(loop.data.linearsolvers), "final_cost": loop.data.bestcost, "it_per_s": a.iters / dt,
                  "mf_trials": st["mf_trials"], "reduced_sweeps": st["reduced_sweeps"], "full_sweeps": st["full_sweeps"]}))
ls.close()
