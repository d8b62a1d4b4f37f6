#!/usr/bin/env python3
"""A/B of accumulate-launch variants INSIDE ONE PROCESS: one context per variant (environment switches are read at upload), the variants
measured in alternating rounds -- in situ (an LM trial between two launches) and back to back -- so that box-to-box and minute-to-minute
drift cancels.  usage: sweep_ab.py NAME=ENV1=V,ENV2=V ...   (NAME= alone: no switches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
variants = []
for a in sys.argv[1:]:
    name, _, envs = a.partition("=")
    variants.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(1000, 100000, 0.01, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
ctxs = []
for name, env in variants:
    for k, v in env.items(): os.environ[k] = v
    c = _capi.Context()
    c.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), int(env.get("FLAGS", "0")))
    for k in env: os.environ.pop(k, None)
    c.set_variables(p.variables); c.sweep_gradhess(); c.damp(1e-3 * c.max_abs_diag())
    ctxs.append(c)
def measure(c, between, n=10):
    c.profile_sweep(True)
    for _ in range(n):
        between(c); c.sweep_gradhess(want_cost=False)
    torch.cuda.synchronize()
    return 1e3 * c.profile_sweep(False, read=True)[0]
res = {name: {"insitu": [], "b2b": [], "trial": []} for name, _ in variants}
import time
for rnd in range(6):
    for (name, _), c in zip(variants, ctxs):
        res[name]["b2b"].append(measure(c, lambda c: None))
        res[name]["insitu"].append(measure(c, lambda c: c.lm_trial(0.0)))
        t0 = time.perf_counter()
        for _ in range(20): c.lm_trial(0.0)
        res[name]["trial"].append(1e6 * (time.perf_counter() - t0) / 20)
for name, r in res.items():
    f = lambda v: f"{np.median(v):6.1f} (" + " ".join(f"{x:.1f}" for x in v) + ")"
    print(f"{name:14s} in situ {f(r['insitu'])}   back to back {f(r['b2b'])}   trial us {f(r['trial'])}", flush=True)
