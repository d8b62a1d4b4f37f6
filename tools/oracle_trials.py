#!/usr/bin/env python3
"""The CPU oracle's own Levenberg-Marquardt loop (oracle/nlls_oracle.c: src/optimize.jl:109-180 + src/iterators.jl:139-172, the reference's full sparse LDL') on bench.py's
workloads: how many damped solves (trials) K iterations take and where it ends -- the figures bench.py prints beside the device's as `lm.oracle_trials` / `lm.oracle_final_cost`.
CPU only (no GPU, no reference needed); writes tests/golden/oracle_trials.json.  Usage: python tools/oracle_trials.py [workload ...]   (about a minute per BASELINE workload)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic
from oracle import oracle as O

CONFIGS = {"ba_100x10k": (100, 10000, 0.1), "ba_1kx100k": (1000, 100000, 0.01), "ba_so3_500x50k": (500, 50000, 0.02)}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oracle_trials.json")


def problem_of(w):          # exactly bench.py's construction
    ncam, npts, prop = CONFIGS[w]
    if w == "ba_so3_500x50k":
        p = synthetic.create_so3_ba_problem(ncam, npts, prop, seed=1, adaptive=True)
    else:
        p = synthetic.create_ba_problem(ncam, npts, prop, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05)
    return synthetic.perturb_ba_problem(p, 1e-3, 1e-3)


def main():
    rec = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for w in (sys.argv[1:] or list(CONFIGS)):
        p = problem_of(w); t0 = time.time()
        op = O.OracleProblem(p.var_kind, p.var_dim, p.groups()); op.set_variables(p.variables)
        entry = {}
        for iters in (20, 25):        # bench.py times 20 iterations behind 5 warm-up iterations of the same start: both counts
            op.set_variables(p.variables)
            r = op.optimize(maxiters=iters, reldcost=(1e300 if w == "ba_so3_500x50k" else -1e300), absdcost=-1e300, dstep=-1.0, maxfails=10 ** 9, maxtime=1e6)
            entry[f"iterations_{iters}"] = {"trials": int(r.linearsolvers), "final_cost": float(r.bestcost), "start_cost": float(r.startcost)}
        entry["what"] = "oracle (CPU restatement of the reference's optimize!, full sparse LDL'): damped solves taken by the first K LM iterations from bench.py's start point, and the best cost reached"
        entry["seconds"] = round(time.time() - t0, 1)
        rec[w] = entry; print(w, json.dumps(entry), flush=True)
        json.dump(rec, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
