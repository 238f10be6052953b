#!/bin/bash
# rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of one bench configuration (GPU box).
# $1 = tag, rest = bench.py arguments.  Output: gpurun_out/prof_$1/
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --steps 60 --warmup 12 --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
echo done $TAG
