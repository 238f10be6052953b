#!/bin/bash
# Lid-driven cavity on small lattices, pull and push.  Output: gpurun_out/bench_small_lid.jsonl
set -u
OUT=gpurun_out/bench_small_lid.jsonl
: > $OUT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/b.log 2>&1 || { echo "FAIL $*"; tail -n 5 gpurun_out/b.log; exit 1; }; tail -n 1 gpurun_out/b.log >> $OUT; }
for n in 512 1024 2048 4096; do
  for scheme in pull push; do
    run --nx $n --ny-per-gpu $n --steps $((2000000 / n)) --warmup $((200000 / n)) --scene lid --scheme $scheme
    run --nx $n --ny-per-gpu $n --steps $((2000000 / n)) --warmup $((200000 / n)) --scene shear --scheme $scheme
  done
done
echo done
