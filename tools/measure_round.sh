#!/bin/bash
# usage (on the GPU box, through gpurun): tools/measure_round.sh <tag>
# bench line with the CPU baseline, rocprofv3 kernel stats of the same command, and the two PMC passes over the sweep
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_stats.err || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python tools/sweep_only.py --reps 3 > /dev/null 2> gpurun_out/${tag}_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python tools/sweep_only.py --reps 3 > /dev/null 2> gpurun_out/${tag}_write.err || exit 4
python tools/sweep_only.py --reps 20 > gpurun_out/${tag}_sweep.json
echo done
