for kv in "A=1" "NLLS_TSP_CAP=1" "NLLS_TSP_CAP=4" "NLLS_TSP_QUAD_MAX=100000" "NLLS_TSP_QUAD_MAX=0" "NLLS_TSP_SCHEME=3" "NLLS_TSP_SCHEME=2" "NLLS_TSP_SLOTS=512" "NLLS_TSP_CAP=1 NLLS_TSP_QUAD_MAX=100000"; do
  echo "== $kv"; env $kv python tools/tsp_try.py --grids 40x40,100x100 --no-dense 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  ', d['grid'], d['default']['reduced_solve_ms'], d['x_relerr_default'])
"
done
