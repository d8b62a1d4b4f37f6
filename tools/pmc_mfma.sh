#!/bin/bash
# usage (on the GPU box, through gpurun): tools/pmc_mfma.sh <tag>
# hardware counters under the matrix-core figures of roofline_solve: f64 MFMA instructions issued, matrix-pipe busy cycles and shader busy
# cycles per kernel of the reduced solve -- the default (block cyclic reduction) and the dense (NLLS_FLAG_NO_BAND) solver, one --pmc pass each
# (the program directly after `--`; no trace domains beside --pmc).  tools/pmc_mfma.py turns the CSVs into profiles/pmc_mfma.json.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/${tag}_mfma_band -- python tools/solve_only.py --reps 4 > gpurun_out/${tag}_mfma_band.json 2> gpurun_out/${tag}_mfma_band.err || exit 3
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/${tag}_mfma_dense -- python tools/solve_only.py --reps 2 --flags 8 > gpurun_out/${tag}_mfma_dense.json 2> gpurun_out/${tag}_mfma_dense.err || exit 4
python tools/pmc_mfma.py ${tag} || exit 5
