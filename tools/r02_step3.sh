#!/bin/bash
set -u
OUT=gpurun_out/r02_step3
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_two_d_cart_gpu.py tests/test_cli.py -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1
rc=$?; tail -4 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
cd build
for g in 1 2 4; do
  for ar in unrolled contracted; do
    LBM_ARITHMETIC=$ar timeout -k 10 120 ./main -nx 16384 -ny 16384 -s 100 -of 0 -gpus $g 2> ../$OUT/main_g${g}_$ar.err || exit 1
    echo "gpus=$g $ar: $(grep 'time loop' ../$OUT/main_g${g}_$ar.err | cut -c1-120)"
  done
done
LBM_ARITHMETIC=contracted timeout -k 10 120 ./main -nx 32768 -ny 32768 -s 100 -of 0 -gpus 8 2> ../$OUT/main_32k_g8.err || exit 1
echo "32768^2 8 slabs on one device: $(grep 'time loop' ../$OUT/main_32k_g8.err | cut -c1-120)"
LBM_ARITHMETIC=contracted timeout -k 10 120 ./main -nx 32768 -ny 32768 -s 100 -of 0 2> ../$OUT/main_32k_g1.err || exit 1
echo "32768^2 single: $(grep 'time loop' ../$OUT/main_32k_g1.err | cut -c1-120)"
