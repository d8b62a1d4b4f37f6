#!/usr/bin/env python3
"""optimizesingles! of all points of a BA workload (cameras fixed): a target for rocprofv3 / quick timing."""
import argparse, os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, kinds as K, _capi
ap = argparse.ArgumentParser(); ap.add_argument("--ncam", type=int, default=1000); ap.add_argument("--npts", type=int, default=100000); ap.add_argument("--prop", type=float, default=0.01)
a = ap.parse_args()
p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(a.ncam, a.npts, a.prop, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 3e-3, 0.0)
pts = np.nonzero((p.var_kind == K.VAR_EUCLIDEAN) & (p.var_dim == 3))[0] + 1
cptr, cgroup, cindex, cslot = p.costlists(pts)
ctx = _capi.Context(0)
ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), 0)
ctx.set_variables(p.variables); c0 = ctx.sweep_cost()
t0 = time.perf_counter(); iters = ctx.optimize_singles(pts, cptr, cgroup, cindex, cslot); t1 = time.perf_counter()
c1 = ctx.sweep_cost()
print(json.dumps({"npoints": int(pts.size), "nblocks": int(cptr[-1]), "call_ms": 1e3 * (t1 - t0), "iters_mean": float(iters.mean()), "iters_max": int(iters.max()), "cost_before": c0, "cost_after": c1}))
