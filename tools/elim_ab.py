#!/usr/bin/env python3
"""A/B of the elimination's member loop: NLLS_ELIM_DMA=0 / 1 (set before the first solve of the process).  Prints the solve time of BASELINE config 4
and checks x against the other variant through a file."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
if wl == "c5":
    p = synthetic.perturb_ba_problem(synthetic.create_so3_ba_problem(500, 50000, 0.02, seed=1, adaptive=True), 1e-3, 1e-3)
else:
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(1000, 100000, 0.01, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
ctx = _capi.Context(0); ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), 0)
ctx.set_variables(p.variables); ctx.sweep_gradhess(); ctx.damp(1e-3 * ctx.max_abs_diag())
x = ctx.solve(want_x=True).copy()
c = ctx.lm_trial(0.0)
ms = ctx.time_solve(20)
tag = os.environ.get("NLLS_ELIM_DMA", "0") + os.environ.get("NLLS_ELIM_NO_FOLD", "0")
np.save(f"/tmp/elim_x_{wl}_{tag}.npy", x)
other = f"/tmp/elim_x_{wl}_{os.environ.get('OTHER', '00')}.npy"
diff = None
if os.path.exists(other):
    xo = np.load(other); diff = float(np.max(np.abs(x - xo)) / np.max(np.abs(xo)))
st = ctx.solve_stats()
print(json.dumps({"workload": wl, "tag": tag, "supernodes": st["elim_supernodes"], "solve_ms": ms, "trial_cost": c, "x_rel_diff_vs_other": diff}))
