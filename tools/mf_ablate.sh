#!/bin/bash
# usage (GPU box): tools/mf_ablate.sh "<dbg values>"  -- mf_elim_kernel's average duration with parts switched off (NLLS_MF_DBG bits: 1 no matrix-core loop, 2 no inverses, 4 no slab writes, 8 no wide supernodes, 16 no merge/flush)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in $1; do
  NLLS_MF_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$d -- python tools/trial_only.py --reps 20 > gpurun_out/abl_$d.json 2> gpurun_out/abl_$d.err
  python - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/abl_$d/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "mf_" in r["Name"]: print("dbg $d", r["Name"][:40].ljust(40), r["Calls"].rjust(4), "%8.1f us" % (float(r["AverageNs"])/1e3))
PY
done
