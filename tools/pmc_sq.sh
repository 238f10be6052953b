#!/bin/bash
# SQ counters of one bench configuration: $1 = tag, rest = bench.py arguments
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline "$@" > $OUT/sq2.log 2>&1 || echo "second pass failed"
echo done
