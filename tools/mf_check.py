"""Matrix-free LM trial against the materialised one on ONE upload (nlls_set_option), and both against the oracle: step, trial cost, statistics, then a short LM loop each way."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
from oracle import oracle as O


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def one(ncam, npts, prop, seed, robust=None, oracle=True):
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=seed, robust=robust, outlier_frac=0.05 if robust else 0.0, outlier_sigma=0.05), 1e-3, 1e-3)
    bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    ctx = _capi.Context(0)
    info = ctx.upload(p.var_kind, p.var_dim, bi, p.groups())
    ctx.set_variables(p.variables)
    c0 = ctx.sweep_gradhess()
    lam = ctx.max_abs_diag() * 1e-6
    out = {}
    for name, mat in (("materialised", 1), ("matrix-free", 0)):
        ctx.set_option(_capi.OPT_MATERIALIZE, mat)
        ctx.set_variables(p.variables); ctx.sweep_gradhess()
        ctx.set_variables(np.zeros(info.var_storage), _capi.VARS_NEXT)
        ct = ctx.lm_trial(lam)
        out[name] = dict(cost=ct, x=ctx.get_step(), v=ctx.get_variables(_capi.VARS_NEXT), q=ctx.quadform(), mx=ctx.step_maxabs(), nrm=ctx.step_norm(), st=ctx.solve_stats())
    a, b = out["materialised"], out["matrix-free"]
    print(f"{ncam}x{npts}: solve_mode {info.solve_mode} nred {info.nreduced_dof} bw {info.bandwidth}; mf trials {b['st']['mf_trials']} (materialised run: {a['st']['mf_trials']}), reduced sweeps {b['st']['reduced_sweeps']}, full {b['st']['full_sweeps']}")
    print(f"   trial cost {a['cost']:.15e} / {b['cost']:.15e}; x rel {rel(b['x'], a['x']):.2e}; vars rel {rel(b['v'], a['v']):.2e}; xHx {a['q'][0]:.12e} / {b['q'][0]:.12e}; gx {a['q'][1]:.12e} / {b['q'][1]:.12e}; max {a['mx']:.6e} / {b['mx']:.6e}")
    if oracle:
        op = O.OracleProblem(p.var_kind, p.var_dim, p.groups()); op.set_variables(p.variables)
        ols = op.linear_system(bi); ols.costgradhess(); ols.solve(lam)
        print(f"   vs oracle: x rel materialised {rel(a['x'], ols.x):.2e}, matrix-free {rel(b['x'], ols.x):.2e}")
    # the gradient on demand after a matrix-free trial
    ctx.set_option(_capi.OPT_MATERIALIZE, 0)
    g_mf = ctx.get_grad(); A_mf = ctx.get_bsm_data()
    ctx.set_option(_capi.OPT_MATERIALIZE, 1); ctx.set_variables(p.variables); ctx.sweep_gradhess()
    print(f"   on-demand b rel {rel(g_mf, ctx.get_grad()):.2e}, A.data rel {rel(A_mf, ctx.get_bsm_data()):.2e}")
    # LM loops
    for name, mat in (("materialised", 1), ("matrix-free", 0)):
        ctx.close()
        q = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=seed, robust=robust, outlier_frac=0.05 if robust else 0.0, outlier_sigma=0.05), 1e-3, 1e-3)
        import os
        os.environ["NLLS_MATERIALIZE"] = str(mat)
        t0 = time.perf_counter(); r = N.optimize(q, N.NLLSOptions(maxiters=15)); dt = time.perf_counter() - t0
        print(f"   optimize ({name}): {r.niterations} iterations, cost {r.startcost:.6e} -> {r.bestcost:.15e}, {r.linearsolvers if hasattr(r, 'linearsolvers') else '?'} solves, {dt * 1e3:.1f} ms")
        ctx = _capi.Context(0)
    ctx.close()


if __name__ == "__main__":
    one(120, 3000, 0.06, 2)
    one(100, 10000, 0.1, 3, robust=N.HuberKernel(0.01))
    one(300, 20000, 0.03, 5, robust=N.HuberKernel(0.01), oracle=False)
