#!/bin/bash
# cfg2 through the command-line driver with and without the periodic code in the bulk
cd build
for rep in 1 2; do
for b in 1 0; do
for a in unrolled contracted; do
  echo "== LBM_FUSED_BULK=$b $a"
  LBM_FUSED_BULK=$b LBM_ARITHMETIC=$a timeout -k 10 100 ./main -ld -nx 1024 -ny 1024 -u 0.1 -s 10000 2>&1 >/dev/null | grep "time loop" | cut -c1-110
done
done
done
echo "== 16384^2 strict push cavity, 1000 steps"
for b in 1 0; do LBM_FUSED_BULK=$b timeout -k 10 100 ./main -ld -nx 16384 -ny 16384 -u 0.1 -s 700 2>&1 >/dev/null | grep "time loop" | cut -c1-110; done
