#!/bin/bash
# SQ instruction counters of the fp64 lid / periodic harness kernels (one pass)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_lid64
mkdir -p $OUT
cd tools
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- ./tune_lid64_wall 8192 8192 > $OUT/sq.log 2>&1 || exit 1
echo done
