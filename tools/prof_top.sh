#!/bin/bash
# usage (on the GPU box): tools/prof_top.sh <tag> [bench args]   -- rocprofv3 kernel stats of a short bench run, top kernels printed
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err
python - <<PY
import csv,glob,json
f=sorted(glob.glob("gpurun_out/$tag/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:${TOPN:-10}]: print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "%9.1f %8s %8s" % (float(r["AverageNs"])/1e3, r["MinNs"], r["MaxNs"]))
for l in open("gpurun_out/$tag.json"):
    if l.startswith("{"):
        d=json.loads(l); print("iters/s", d["value"], "trials/s", d["lm"]["lm_trials_per_s"], "solves", d["lm"]["linear_solves"], "solve_ms", d["roofline"]["solve_ms"])
PY
