#!/bin/bash
# usage (on the GPU box, through gpurun): tools/pmc_trial.sh <tag>  -- issue counters per kernel of the LM trial (tools/trial_only.py): vector / scalar / LDS / matrix-core instructions,
# busy cycles and waves, one --pmc pass (nothing beside --pmc); prints per kernel: calls, instructions per call
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/${tag}_trial_pmc -- python tools/trial_only.py --reps 10 > gpurun_out/${tag}_trial_pmc.json 2> gpurun_out/${tag}_trial_pmc.err || exit 3
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${tag}_trial_pmc2 -- python tools/trial_only.py --reps 10 > /dev/null 2> gpurun_out/${tag}_trial_pmc2.err || exit 4
python - <<PY
import csv,glob,collections
for d in ("gpurun_out/${tag}_trial_pmc","gpurun_out/${tag}_trial_pmc2"):
    f=sorted(glob.glob(d+"/*/*counter_collection.csv"))[-1]
    acc=collections.defaultdict(lambda: collections.Counter()); calls=collections.Counter(); seen=set()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void nlls::","").replace("nlls::","")[:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); calls[k]+=1
    for k,c in sorted(acc.items(), key=lambda kv:-sum(kv[1].values()))[:8]:
        print(k.ljust(40), calls[k], {n: round(v/max(calls[k],1)) for n,v in c.items()})
PY
