#!/bin/bash
OUT=gpurun_out/$1
: > $OUT
m() { timeout -k 5 60 tools/micro/march "$@" >> $OUT 2>&1 || { echo "failed: $*" >> $OUT; exit 1; }; }
N="16384 16384"
#      LB RB W  H seg  grp wpc extra barrier
for S in 16 32 64 128 256 512; do m $N 8 1 56 4 $S 4 8 0 0; done
for S in 32 64 128 256; do m $N 8 1 56 4 $S 4 8 0 1; done
m $N 8 1 56 4 128 8 8 0 0
m $N 8 1 56 4 128 8 8 0 1
m $N 8 1 56 4 128 2 8 0 1
m $N 8 2 56 4 128 4 8 0 1
m $N 16 1 56 4 128 4 8 0 1
m $N 16 1 56 4 128 4 4 0 1
m $N 16 1 56 4 64 4 8 0 0
m $N 16 1 56 4 32 4 8 0 0
cat $OUT
