#!/bin/bash
# rocprofv3 passes for the headline bench command (GPU box). Each pass is its own run.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r01
mkdir -p $OUT
BENCH="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace exit $?"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch exit $?"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1; echo "write exit $?"
find $OUT -name "*.csv" | head -20
