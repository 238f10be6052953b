#!/bin/bash
# Bench lines for every sweep flavour at 16384^2 (fused time loop and one-step), the slab path and two
# smaller lattices.  Output: gpurun_out/bench_matrix_fused.jsonl
set -u
OUT=gpurun_out/bench_matrix_fused.jsonl
: > $OUT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/b.log 2>&1 || { echo "FAIL $*"; tail -n 5 gpurun_out/b.log; exit 1; }; tail -n 1 gpurun_out/b.log >> $OUT; }
for prec in fp32 fp64; do
  for flav in "--scene shear --scheme pull" "--scene shear --scheme push" "--scene lid --scheme pull" "--scene lid --scheme push"; do
    run --precision $prec $flav
    run --precision $prec $flav --max-fuse 1
  done
done
run --slab
run --slab --scheme push
run --slab --precision fp64
run --slab --scene lid --scheme push
run --nx 1024 --ny-per-gpu 1024 --steps 2000 --warmup 200
run --nx 1024 --ny-per-gpu 1024 --steps 2000 --warmup 200 --max-fuse 1
run --nx 4096 --ny-per-gpu 4096 --steps 400 --warmup 40
run --nx 22627 --ny-per-gpu 22627 --precision fp64 --steps 40 --warmup 8
run --nx 22627 --ny-per-gpu 22627 --precision fp64 --steps 40 --warmup 8 --max-fuse 1
run --nx 32768 --ny-per-gpu 32768 --steps 40 --warmup 8
run --nx 8192 --ny-per-gpu 8192 --steps 200 --warmup 20
run --nx 2048 --ny-per-gpu 2048 --steps 1000 --warmup 100
run --nx 512 --ny-per-gpu 512 --steps 3000 --warmup 300
run --nx 1024 --ny-per-gpu 1024 --steps 2000 --warmup 200 --scene lid --scheme push
for prec in fp32 fp64; do
  run --layout aos --precision $prec
  run --layout aos --precision $prec --scheme push
  run --layout aos --precision $prec --scene lid --scheme pull
  run --layout aos --precision $prec --scene lid --scheme push
done
echo done
