#!/bin/bash
# Run on the MI355X box (via gpurun): kernel-trace stats + HBM traffic counters for bench.py.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Outputs land in gpurun_out/$1/.
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
python3 tools/pmc_summary.py $OUT
