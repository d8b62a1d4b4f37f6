#!/bin/bash
# usage (on the GPU box): tools/tsp_prof.sh <tag> [tsp_try args]   -- rocprofv3 kernel stats of the tile-sparse reduced solve on a camera grid
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python tools/tsp_try.py "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err
python - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/$tag/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:${TOPN:-14}]: print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "%9.1f %8s %8s" % (float(r["AverageNs"])/1e3, r["MinNs"], r["MaxNs"]))
PY
cat gpurun_out/$tag.json
