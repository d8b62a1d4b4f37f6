#!/bin/bash
# usage (gpurun): tools/piece_prof.sh <workload> <piece> ...   -- kernel-trace averages of the solve's kernels with runs of eliminated blocks cut into pieces of <piece> members (NLLS_SUPERNODE_PIECE)
w=$1; shift
cd /tmp; export TMPDIR=/tmp
for p in "$@"; do
  NLLS_SUPERNODE_PIECE=$p rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/piece_${w}_$p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/piece_${w}_$p.log 2> $GRAFT_REPO_ROOT/gpurun_out/piece_${w}_$p.err
  f=$(ls $GRAFT_REPO_ROOT/gpurun_out/piece_${w}_$p/*/*kernel_stats.csv | head -1)
  echo "piece $p: $(python3 -c "
import csv,sys,json
rows={r['Name']:float(r['AverageNs'])/1e3 for r in csv.DictReader(open('$f'))}
print({s:round(v,1) for k,v in rows.items() for s in ('schur_elim_all','backsub_fast','gh_f','cost_kernel<1, true','cost_kernel<8, true') if s in k})
d=json.loads([l for l in open('$GRAFT_REPO_ROOT/gpurun_out/piece_${w}_$p.log') if l.startswith('{')][-1]); print(d['value'], d['roofline']['solve_ms'])
")"
done
