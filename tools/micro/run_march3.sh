#!/bin/bash
OUT=gpurun_out/$1
: > $OUT
m() { timeout -k 5 60 tools/micro/march "$@" >> $OUT 2>&1 || { echo "failed: $*" >> $OUT; exit 1; }; }
N="16384 16384"
#      LB RB W  H seg  grp wpc extra barrier layout
for S in 32 64 118 128 186 256; do m $N 8 1 56 4 $S 4 8 0 0 1; done
m $N 8 1 56 4 118 4 8 12 0 1
m $N 8 1 56 4 118 1 8 0 0 1
m $N 8 1 56 4 118 4 4 0 0 1
m $N 8 1 56 4 118 4 12 0 0 1
m $N 8 2 56 4 118 4 8 0 0 1
m $N 16 1 56 4 118 4 8 0 0 1
m $N 16 1 56 4 118 4 4 0 0 1
m $N 8 1 64 0 128 4 8 0 0 1
m 16640 16384 8 1 56 4 118 4 8 0 0 1
m 16384 16384 8 1 56 4 118 4 8 0 0 0
cat $OUT
