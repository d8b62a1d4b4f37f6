#!/bin/bash
# usage (on the GPU box): tools/pmc_quick.sh <tag>  -- FETCH_SIZE / WRITE_SIZE of the accumulate launch (two passes), bytes per launch printed
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python tools/sweep_only.py --reps 3 > /dev/null 2> gpurun_out/${tag}_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python tools/sweep_only.py --reps 3 > /dev/null 2> gpurun_out/${tag}_write.err || exit 4
python - <<PY
import csv,glob
def per(pattern, counter):
    f=sorted(glob.glob(pattern))[-1]
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]==counter and "gh_fused" in r["Kernel_Name"]]
    return sum(v)/len(v)*1024
fe=per("gpurun_out/${tag}_fetch/*/*counter_collection.csv","FETCH_SIZE"); wr=per("gpurun_out/${tag}_write/*/*counter_collection.csv","WRITE_SIZE")
print("fetch (x2) %.1f MB  write %.1f MB  total %.1f MB  = %.3f x 188.384 MB" % (2*fe/1e6, wr/1e6, (2*fe+wr)/1e6, (2*fe+wr)/188.384e6))
PY
