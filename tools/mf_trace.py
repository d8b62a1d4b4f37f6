"""Cost per LM iteration, matrix-free against materialised (and the deterministic materialised assembly), one problem."""
import sys, os
import numpy as np
sys.path.insert(0, ".")
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi, iterators as It, optimizer as Opt
from nllssolver_jl_amd.dist import ShardedLS
import time
ncam, npts, prop, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
niter = int(sys.argv[5]) if len(sys.argv) > 5 else 15
def run(mat, flags=0):
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=seed, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
    ls = ShardedLS(p, np.ones(p.nvariables, bool), flags=flags, device=0, rank=0, world=1, dist=None, host_staged=False)
    ls.ctx.set_option(_capi.OPT_MATERIALIZE, mat)
    options = N.NLLSOptions(maxiters=10 ** 9, reldcost=-np.inf, absdcost=-np.inf, dstep=-1.0, maxfails=10 ** 9, maxtime=1e6)
    ls.ctx.set_variables(p.variables, _capi.VARS_CURRENT); ls.ctx.copy_variables(_capi.VARS_NEXT, _capi.VARS_CURRENT)
    data = Opt.NLLSInternal(ls, time.perf_counter_ns())
    loop = Opt.OuterLoop(p, options, data, It.LevMarData(), It.iterate_levmar, N.nullcallback); loop.start()
    out = []
    for it in range(niter):
        s0 = loop.data.linearsolvers; loop.iterations(1); out.append((loop.data.bestcost, loop.data.linearsolvers - s0, loop.iterdata.lambda_ if hasattr(loop, "iterdata") else 0))
    ls.close()
    return out
a = run(1); b = run(0); c = run(1, _capi.FLAG_DETERMINISTIC)
for i, (x, y, z) in enumerate(zip(a, b, c)):
    print(f"{i+1:3d} mat {x[0]:.15e} ({x[1]})  mf {y[0]:.15e} ({y[1]})  det {z[0]:.15e} ({z[1]})  rel mf-mat {abs(y[0]-x[0])/abs(x[0]):.1e} det-mat {abs(z[0]-x[0])/abs(x[0]):.1e}")
