#!/usr/bin/env python3
"""Runs only the accumulate (gradient) sweep and the cost sweep of a BA workload a few times: a short target for
rocprofv3 --pmc passes (HBM traffic, LDS conflicts) and quick A/B timing of the sweep kernels."""
import argparse, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi

ap = argparse.ArgumentParser()
ap.add_argument("--ncam", type=int, default=1000); ap.add_argument("--npts", type=int, default=100000)
ap.add_argument("--prop", type=float, default=0.01); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--workload", default=None, help="a bench.py workload name (ba_100x10k, ba_1kx100k, ba_so3_500x50k) instead of --ncam/--npts/--prop")
a = ap.parse_args()
if a.workload == "ba_so3_500x50k":
    p = synthetic.perturb_ba_problem(synthetic.create_so3_ba_problem(500, 50000, 0.02, seed=1, adaptive=True), 1e-3, 1e-3)
else:
    if a.workload:
        a.ncam, a.npts, a.prop = {"ba_100x10k": (100, 10000, 0.1), "ba_1kx100k": (1000, 100000, 0.01)}[a.workload]
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(a.ncam, a.npts, a.prop, seed=1, robust=N.HuberKernel(0.01),
                                                                 outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
ctx = _capi.Context(0)
info = ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), a.flags)
ctx.set_variables(p.variables)
ms = ctx.time_sweep_accumulate(a.reps); msg = ctx.time_sweep_gradhess(a.reps); msc = ctx.time_sweep_cost(a.reps)
nobs = p.ncosts()
from nllssolver_jl_amd import kinds as K
alg = sum(len(g) * 8 * (K.res_ndata(g.res_kind) + K.res_ndeps(g.res_kind)) for g in p.costs.values()) + 8 * info.var_storage + 8 * (info.nnz_data + info.ndof)
print(json.dumps({"nobs": nobs, "sweep_ms": ms, "sweep_with_cost_ms": msg, "cost_ms": msc, "alg_bytes": alg, "GBps": alg / ms / 1e6, "owner_path": info.owner_path}))
