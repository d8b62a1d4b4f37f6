#!/bin/bash
# usage (on the GPU box, through gpurun): tools/pmc_iter.sh <tag> [workload]
# HBM bytes of ONE LM iteration, matrix-free and materialised: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, nothing beside --pmc) over tools/lm_iters.py,
# every kernel of the loop counted (the upload's own kernels -- there are none -- and the first full sweep included: 20 iterations dilute them); tools/pmc_iter.py -> profiles/pmc_iter.json
tag=$1; w=${2:-ba_1kx100k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for path in mf mat; do
  flag=$([ $path = mat ] && echo "--materialise" || echo "")
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_iter_fetch_${path}_$w -- python tools/lm_iters.py --workload $w $flag > gpurun_out/${tag}_iter_${path}_$w.json 2> gpurun_out/${tag}_iter_fetch_${path}_$w.err || exit 3
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_iter_write_${path}_$w -- python tools/lm_iters.py --workload $w $flag > /dev/null 2> gpurun_out/${tag}_iter_write_${path}_$w.err || exit 4
done
python tools/pmc_iter.py $tag $w
