#!/bin/bash
# usage: tools/prof_stats.sh <outname> <python script + args...>   (on the GPU box, through gpurun)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -- python "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/$out/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r['Percentage']}")
PY
