"""Reduced solve of camera chains from 50 to 1000 cameras (300 .. 6000 reduced dof): block cyclic reduction (default) against the one-workgroup chain kernels
(NLLS_FLAG_NO_BCR).  Prints, per size: (reduced solve us, whole solve us, solve_mode, bandwidth) for both.  gpurun -- python tools/chain_vs_bcr.py   """
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
for ncam in (50, 100, 150, 200, 300, 400, 600, 1000):
    npts = ncam * 100; prop = 10.0 / ncam
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
    out = {}
    for name, flags in (("bcr", 0), ("chain", _capi.FLAG_NO_BCR)):
        ctx = _capi.Context(0)
        info = ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), flags)
        ctx.set_variables(p.variables); ctx.sweep_gradhess(); ctx.damp(1e-3 * ctx.max_abs_diag())
        ctx.time_solve(2)
        out[name] = (round(1e3 * ctx.time_reduced_solve(10), 1), round(1e3 * ctx.time_solve(10), 1), int(info.solve_mode), int(info.bandwidth))
        ctx.close()
    print(ncam, 6 * ncam, out, flush=True)
