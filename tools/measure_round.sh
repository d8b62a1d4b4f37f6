#!/bin/bash
# usage (on the GPU box, through gpurun): tools/measure_round.sh <tag>
# the two PMC passes over the sweep first (-> profiles/pmc_traffic.json with the hash of the sweep sources, so that the bench line carries `traffic`),
# then the bench line with the CPU baseline (the driver's configuration: --steps 20 --warmup 5), rocprofv3 kernel stats of the same command, and the bench lines of the other BASELINE configurations
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_stats.err || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python tools/sweep_only.py --reps 3 > /dev/null 2> gpurun_out/${tag}_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python tools/sweep_only.py --reps 3 > /dev/null 2> gpurun_out/${tag}_write.err || exit 4
python tools/pmc_traffic.py ${tag} > gpurun_out/${tag}_traffic.json || exit 5
# ... the same two passes for BASELINE configs 3 and 5, and one pass of issue counters (vector / matrix instructions, busy cycles) over their sweeps: "compute-bound" with a counter
for w in ba_100x10k ba_so3_500x50k; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch_$w -- python tools/sweep_only.py --workload $w --reps 3 > /dev/null 2> gpurun_out/${tag}_fetch_$w.err || exit 11
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write_$w -- python tools/sweep_only.py --workload $w --reps 3 > /dev/null 2> gpurun_out/${tag}_write_$w.err || exit 12
  python tools/pmc_traffic.py ${tag} $w > gpurun_out/${tag}_traffic_$w.json || exit 13
  rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_valu_$w -- python tools/sweep_only.py --workload $w --reps 3 > /dev/null 2> gpurun_out/${tag}_valu_$w.err || exit 14
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_valu_ba_1kx100k -- python tools/sweep_only.py --reps 3 > /dev/null 2> gpurun_out/${tag}_valu_ba_1kx100k.err || exit 14
cp profiles/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
# hardware counters under the matrix-core figures (f64 MFMA instructions, matrix-pipe busy cycles): profiles/pmc_mfma.json, checked by bench.py against the launcher's count
bash tools/pmc_mfma.sh ${tag} > gpurun_out/${tag}_pmc_mfma.txt 2>&1 || exit 10
cp profiles/pmc_mfma.json gpurun_out/${tag}_pmc_mfma.json
cp profiles/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
for w in ba_100x10k curvefit_10k ba_so3_500x50k ba_10kx1M; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || exit 6; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --shuffle-cameras 7 > gpurun_out/${tag}_bench_shuffled.json 2> gpurun_out/${tag}_bench_shuffled.err || exit 15
python bench.py --solver dense --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_dense.json 2> gpurun_out/${tag}_bench_dense.err || exit 7
# rocprofv3 kernel stats of the other BASELINE configurations (2, 3, 5) and of the dense solver
for w in ba_100x10k curvefit_10k ba_so3_500x50k; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$w -- python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_stats_$w.err || exit 8; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_dense -- python bench.py --solver dense --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_stats_dense.err || exit 9
# the tile-sparse reduced solver (solve_mode 3): camera grids -- LM lines, kernel stats of the 40 x 40 grid, the reduced solve against the windowed and the dense LDL'
for w in ba_grid_40x40 ba_grid_100x100; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || exit 16; done
python bench.py --workload ba_grid_40x40 --solver windowed --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_ba_grid_40x40_windowed.json 2> gpurun_out/${tag}_bench_ba_grid_40x40_windowed.err || exit 17
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_ba_grid_40x40 -- python bench.py --workload ba_grid_40x40 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_stats_ba_grid_40x40.err || exit 18
python tools/tsp_try.py --grids 24x24,40x40 > gpurun_out/${tag}_tsp_try.json 2> gpurun_out/${tag}_tsp_try.err || exit 19
python tools/tsp_try.py --grids 100x100 --no-dense >> gpurun_out/${tag}_tsp_try.json 2>> gpurun_out/${tag}_tsp_try.err || exit 20
echo done
