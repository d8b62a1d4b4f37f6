#!/bin/bash
# A/B switches of the tile-sparse reduced solver on two camera grids (run on the GPU box): reduced solve in ms, x against the windowed dense LDL'
for kv in "A=1" "NLLS_TSP_NO_MASKS=1" "NLLS_TSP_CARRY=0" "NLLS_TSP_CARRY=0 NLLS_TSP_NO_MASKS=1" "NLLS_TSP_CARRY=100" "NLLS_TSP_CARRY=48" "NLLS_TSP_CAP=1" "NLLS_TSP_CAP=4" "NLLS_TSP_QUAD_MAX=0" "NLLS_TSP_SCHEME=3" "NLLS_TSP_SCHEME=2"; do
  echo "== $kv"; env $kv python tools/tsp_try.py --grids 24x24,40x40,100x100 --no-dense 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  ', d['grid'], d['default']['reduced_solve_ms'], d['x_relerr_default'])
"
done
