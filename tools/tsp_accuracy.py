"""Residual of the damped step on the device's own (H + lambda I) x = -g for a range of dampings, through the tile-sparse solver and the dense LDL' of the same system
(a 30 x 30 camera grid): what the explicit inverses of the leaf levels' pivot tiles and the atomic pieces of the update cost in accuracy."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
from tests.helpers import bsm_to_csr
p = synthetic.perturb_ba_problem(synthetic.create_grid_ba_problem(30, 30, 4, seed=3, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05, noise=1e-3), 1e-3, 1e-3)
bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
for name, flags in (("tile-sparse", 0), ("windowed", _capi.FLAG_NO_TILE_SPARSE), ("dense", _capi.FLAG_NO_BAND)):
    ctx = _capi.Context(0); info = ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), flags)
    ctx.set_variables(p.variables); ctx.sweep_gradhess()
    H = bsm_to_csr(ctx.bsm_index(), ctx.get_bsm_data(), info.ndof); g = ctx.get_grad(); md = ctx.max_abs_diag()
    out = {}
    for sc in (1e-2, 1e-4, 1e-6, 1e-8, 1e-10, 1e-12):
        lam = sc * md; ctx.sweep_gradhess(); ctx.damp(lam)          # (the sweep resets the damping: nlls_damp adds to what is there)
        try:
            x = ctx.solve(want_x=True); out[f"{sc:g}"] = float(np.linalg.norm(H @ x + lam * x + g) / np.linalg.norm(g))
        except Exception as e:
            out[f"{sc:g}"] = str(e)[:40]
    print(json.dumps({"solver": name, "solve_mode": int(info.solve_mode), "relative residual by lambda / max H_ii": out}))
    ctx.close()
