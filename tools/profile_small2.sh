#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_small2
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --nx 1024 --ny-per-gpu 1024 --steps 400 --warmup 40 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
