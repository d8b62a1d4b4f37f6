#!/bin/bash
# usage (GPU box): tools/mf_variants.sh  -- mf_elim_kernel's duration for each library under nllssolver.jl_amd/csrc/variants/ (A/B builds of nlls_mf.hip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in nllssolver.jl_amd/csrc/variants/*.so; do
  t=$(basename $lib .so)
  for d in 0 1; do
  NLLS_AMD_LIB=$PWD/$lib NLLS_MF_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var_${t}_$d -- python tools/trial_only.py --reps 20 > gpurun_out/var_${t}_$d.json 2> gpurun_out/var_${t}_$d.err
  python - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/var_${t}_$d/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "mf_elim" in r["Name"]: print("$t dbg $d", "%8.1f us" % (float(r["AverageNs"])/1e3))
PY
  done
done
