#!/bin/bash
# usage (on the GPU box, through gpurun): tools/measure_round.sh <tag>
# kernel traces of every BASELINE configuration first (their average dispatch durations go into profiles/pmc_traffic.json beside the PMC bytes), then the two PMC passes over
# the sweep per configuration, the issue counters, the matrix-core counters, and the bench lines (the driver's configuration: --steps 20 --warmup 5)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_stats.err || exit 2
for w in ba_100x10k curvefit_10k ba_so3_500x50k; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$w -- python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_stats_$w.err || exit 8; done
for w in ba_1kx100k ba_100x10k ba_so3_500x50k; do
  sfx=$([ $w = ba_1kx100k ] && echo "" || echo "_$w")
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch$sfx -- python tools/sweep_only.py --workload $w --reps 3 > /dev/null 2> gpurun_out/${tag}_fetch$sfx.err || exit 11
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write$sfx -- python tools/sweep_only.py --workload $w --reps 3 > /dev/null 2> gpurun_out/${tag}_write$sfx.err || exit 12
  python tools/pmc_traffic.py ${tag} $w > gpurun_out/${tag}_traffic$sfx.json || exit 13
  rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_valu_$w -- python tools/sweep_only.py --workload $w --reps 3 > /dev/null 2> gpurun_out/${tag}_valu_$w.err || exit 14
done
cp profiles/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
# HBM bytes of ONE LM iteration, matrix-free and materialised (every kernel of the loop): profiles/pmc_iter.json, quoted by bench.py as hbm_bytes_per_lm_iteration
for w in ba_1kx100k ba_100x10k; do bash tools/pmc_iter.sh ${tag} $w > gpurun_out/${tag}_pmc_iter_$w.txt 2>&1 || exit 19; done
# hardware counters under the matrix-core figures (f64 MFMA instructions, matrix-pipe busy cycles): profiles/pmc_mfma.json, checked by bench.py against the launcher's count
bash tools/pmc_mfma.sh ${tag} > gpurun_out/${tag}_pmc_mfma.txt 2>&1 || exit 10
cp profiles/pmc_mfma.json gpurun_out/${tag}_pmc_mfma.json
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
for w in ba_100x10k curvefit_10k ba_so3_500x50k ba_10kx1M; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || exit 6; done
# the same loop through the MATERIALISING kernels (rounds 1-5: accumulate sweep -> A.data -> elimination), bench line and kernel trace
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --path materialise > gpurun_out/${tag}_bench_materialised.json 2> gpurun_out/${tag}_bench_materialised.err || exit 20
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_materialised -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --path materialise > /dev/null 2> gpurun_out/${tag}_stats_materialised.err || exit 21
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --shuffle-cameras 7 > gpurun_out/${tag}_bench_shuffled.json 2> gpurun_out/${tag}_bench_shuffled.err || exit 15
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --solver nofloor > gpurun_out/${tag}_bench_nofloor.json 2> gpurun_out/${tag}_bench_nofloor.err || exit 16
python bench.py --solver dense --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_dense.json 2> gpurun_out/${tag}_bench_dense.err || exit 7
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_dense -- python bench.py --solver dense --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_stats_dense.err || exit 9
# the tile-sparse reduced solver (solve_mode 3) on a camera grid
python bench.py --workload ba_grid_40x40 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_ba_grid_40x40.json 2> gpurun_out/${tag}_bench_ba_grid_40x40.err || exit 17
NLLS_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_forcedist.json 2> gpurun_out/${tag}_bench_forcedist.err || exit 18
echo done
