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
ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), int(os.environ.get('NLLS_INSITU_FLAGS', '0')))
ctx.set_variables(p.variables)
st = torch.cuda.Stream(); ctx.set_stream(st.cuda_stream)
big = torch.empty(1 << 27, dtype=torch.float64, device="cuda")     # 1 GiB
ctx.sweep_gradhess(); ctx.damp(1e-3 * ctx.max_abs_diag())
ONLY = os.environ.get("NLLS_INSITU_ONLY")
def run(name, between, n=12):
    if ONLY and not any(k in name for k in ONLY.split(",")): return
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
# which part of the solve costs the launch its warm start?
run("elimination only", lambda: ctx.solve_local())
def reduced():
    ctx.time_reduced_solve(1)
run("elimination + 2 reduced", reduced)
run("solve + cost sweep", lambda: (ctx.solve(want_x=False), ctx.sweep_cost(_capi.VARS_CURRENT)))
def trial():
    ctx.lm_trial(0.0)
run("whole LM trial", trial)
med = torch.empty(1 << 23, dtype=torch.float64, device="cuda")     # 64 MiB
def fill64():
    with torch.cuda.stream(st): med.zero_()
run("64 MiB fill", fill64)
def read_big():
    with torch.cuda.stream(st): big[: 1 << 25].sum()               # 256 MiB read
run("256 MiB read", read_big)

# is it the cache, or the chip's clocks after a stretch of small launches?  (the reduced solve touches 7 MB)
tiny = torch.zeros(64, dtype=torch.float64, device="cuda")
def tinies(n):
    def f():
        with torch.cuda.stream(st):
            for _ in range(n): tiny.add_(1.0)
    return f
run("22 tiny launches", tinies(22))
run("60 tiny launches", tinies(60))
run("200 tiny launches", tinies(200))
alu = torch.rand(1 << 21, dtype=torch.float32, device="cuda")
def tiny_then_alu():
    with torch.cuda.stream(st):
        for _ in range(60): tiny.add_(1.0)
        alu.cos_()
run("60 tiny + one 8 MB cos", tiny_then_alu)
import time
def sleepy():
    torch.cuda.synchronize(); time.sleep(0.002)
run("2 ms idle", sleepy)

# how close to the edge of the 256 MB memory-side cache is the launch's footprint (216 MB)?  Fresh data of growing size in between
for mb in (4, 8, 16, 32, 48):
    n8 = mb << 17
    def rd(n8=n8):
        with torch.cuda.stream(st): big[(1 << 26):(1 << 26) + n8].sum()
    run(f"{mb} MiB read of other data", rd)
for mb in (4, 8, 16, 32):
    n8 = mb << 17
    def wr(n8=n8):
        with torch.cuda.stream(st): big[(1 << 26):(1 << 26) + n8].zero_()
    run(f"{mb} MiB fill of other data", wr)
