#!/bin/bash
# cfg3 through the command-line driver, both arithmetics: time-loop line and whole-run line (stderr)
set -e
for a in unrolled contracted; do
  for rep in 1 2 3; do
    echo "== arithmetic=$a rep=$rep"
    LBM_ARITHMETIC=$a timeout -k 10 120 ./build/main -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay 2>&1 >/dev/null | tail -n 3
  done
done
echo "== cfg1 1024^2 100000 steps, no output"
for a in unrolled contracted; do
  LBM_ARITHMETIC=$a timeout -k 10 120 ./build/main -nx 1024 -ny 1024 -s 100000 -of 0 --shear-wave-decay 2>&1 >/dev/null | tail -n 3
done
