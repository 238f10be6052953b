#!/bin/bash
set -u
export TMPDIR=/tmp
for k in 1 2 3; do
OUT=gpurun_out/prof_slab3_$k
mkdir -p $OUT
LBM_BENCH_STREAMS=$k timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --slab > $OUT/trace.log 2>&1 || exit 1
tail -n 1 $OUT/trace.log | cut -c1-60
done
