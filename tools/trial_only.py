"""Repeated Levenberg-Marquardt trials (damp, solve, retract, cost sweep) of BASELINE config 4 on one gradient / Hessian -- for rocprofv3 runs
of the kernels of a trial alone (the LM loop of bench.py interleaves them with the accumulate sweep)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=40); ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--ncam", type=int, default=1000); ap.add_argument("--npts", type=int, default=100000); ap.add_argument("--prop", type=float, default=0.01)
a = ap.parse_args()
p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(a.ncam, a.npts, a.prop, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
ctx = _capi.Context(0)
bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
info = ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), a.flags)
ctx.set_variables(p.variables); ctx.sweep_gradhess(); lam = 1e-4 * ctx.max_abs_diag()
for _ in range(5): ctx.lm_trial(lam)
t0 = time.perf_counter()
for _ in range(a.reps): c = ctx.lm_trial(lam)
dt = time.perf_counter() - t0
print(json.dumps({"us_per_trial": 1e6 * dt / a.reps, "cost": c}))
ctx.close()
