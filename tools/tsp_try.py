"""The tile-sparse reduced solver on camera-grid bundle adjustments: solve mode chosen, x against the dense LDL' of the same system, time of the reduced solve
under the three solvers (tile-sparse / windowed dense / dense)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi

ap = argparse.ArgumentParser(); ap.add_argument("--grids", default="24x24,40x40"); ap.add_argument("--pts", type=int, default=4); ap.add_argument("--shuffle", type=int, default=None)
ap.add_argument("--no-dense", action="store_true")
a = ap.parse_args()
for g in a.grids.split(","):
    gw, gh = (int(v) for v in g.split("x"))
    p = synthetic.create_grid_ba_problem(gw, gh, a.pts, seed=3, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05, noise=1e-3)
    if a.shuffle is not None: p = synthetic.shuffle_camera_labels(p, gw * gh, a.shuffle)
    p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
    bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    xs = {}; out = {"grid": g, "reduced_dof": 6 * gw * gh}
    for name, flags in (("default", 0), ("windowed", _capi.FLAG_NO_TILE_SPARSE), ("dense", _capi.FLAG_NO_BAND)):
        if name == "dense" and a.no_dense: continue
        ctx = _capi.Context(0)
        t0 = time.time(); info = ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), flags); tu = time.time() - t0
        st = ctx.solve_stats()
        ctx.set_variables(p.variables); ctx.sweep_gradhess(); ctx.damp(1e-4 * ctx.max_abs_diag()); ctx.solve()
        xs[name] = ctx.get_step().copy()
        out[name] = {"solve_mode": info.solve_mode, "dense_window": st["dense_window"], "upload_s": round(tu, 2), "reduced_solve_ms": round(ctx.time_reduced_solve(3), 3), "solve_ms": round(ctx.time_solve(3), 3)}
        ctx.close()
    ref = xs.get("dense", xs["windowed"])
    out["x_relerr_default"] = float(np.linalg.norm(xs["default"] - ref) / np.linalg.norm(ref))
    print(json.dumps(out), flush=True)
