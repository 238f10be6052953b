#!/bin/bash
OUT=gpurun_out/$1
: > $OUT
m() { timeout -k 5 60 tools/micro/march "$@" >> $OUT 2>&1 || { echo "failed: $*" >> $OUT; exit 1; }; }
N="16384 16384"
#      LB RB W  H seg  grp wpc extra barrier layout
for S in 1 2 4 8 16; do m $N 8 1 56 4 $S 4 8 0 0 0; done
for G in 4 8 16; do for b in 0 1; do m $N 8 1 56 4 128 $G 16 0 $b 0; done; done
for G in 8 16; do for b in 0 1; do m $N 8 1 56 4 128 $G 8 0 $b 0; done; done
m $N 8 2 56 4 128 8 8 0 1 0
m $N 8 2 56 4 128 16 16 0 1 0
cat $OUT
