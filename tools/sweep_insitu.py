#!/usr/bin/env python3
"""Why is the accumulate launch slower inside the LM loop than back to back?  The launch's execution span (per-workgroup clock stamps,
nlls_profile_sweep) with different things run between two launches: nothing, a 1 GiB fill (cold L2 / MALL), a cost sweep, a solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(1000, 100000, 0.01, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
ctx = _capi.Context()
ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), 0)
ctx.set_variables(p.variables)
st = torch.cuda.Stream(); ctx.set_stream(st.cuda_stream)
big = torch.empty(1 << 27, dtype=torch.float64, device="cuda")     # 1 GiB
ctx.sweep_gradhess(); ctx.damp(1e-3 * ctx.max_abs_diag())
def run(name, between, n=12):
    ctx.profile_sweep(True)
    for _ in range(n):
        between()
        ctx.sweep_gradhess(want_cost=False)
    torch.cuda.synchronize()
    r = ctx.profile_sweep(False, read=True)
    print(f"{name:28s} avg {1e3*r[0]:.1f} us  min {1e3*r[1]:.1f}  max {1e3*r[2]:.1f}  n {int(r[3])}", flush=True)
def fill():
    with torch.cuda.stream(st): big.zero_()
def fill_small():
    with torch.cuda.stream(st): big[: 1 << 24].zero_()       # 128 MiB
run("nothing (back to back)", lambda: None)
run("1 GiB fill", fill)
run("128 MiB fill", fill_small)
run("cost sweep", lambda: ctx.sweep_cost_async(_capi.VARS_CURRENT) if hasattr(ctx, "sweep_cost_async") else ctx.sweep_cost(_capi.VARS_CURRENT))
run("solve", lambda: ctx.solve(want_x=False))
run("synchronize only", lambda: torch.cuda.synchronize())
