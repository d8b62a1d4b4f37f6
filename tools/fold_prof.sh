#!/bin/bash
# usage (GPU box): tools/fold_prof.sh <tag> [workload]   -- kernel trace + LDS / issue counters of the accumulate sweep alone (tools/sweep_only.py)
tag=$1; w=${2:-ba_so3_500x50k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -- python tools/sweep_only.py --workload $w --reps 20 > gpurun_out/${tag}_kt.json 2> gpurun_out/${tag}_kt.err || exit 2
find gpurun_out/${tag}_kt -name "*kernel_stats.csv" | xargs cat | cut -c1-200 | head -8
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/${tag}_pmc -- python tools/sweep_only.py --workload $w --reps 3 > /dev/null 2> gpurun_out/${tag}_pmc.err || exit 3
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/${tag}_pmc/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"][:40]
        if "gh_" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
