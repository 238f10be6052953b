#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_slab2
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --slab > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_plain -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $OUT/trace_plain.log 2>&1 || exit 1
