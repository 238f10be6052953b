#!/bin/bash
# run_march.sh OUT: sweeps of the marching access pattern (tools/micro/march.hip); one line per setting
OUT=gpurun_out/$1
: > $OUT
m() { timeout -k 5 60 tools/micro/march "$@" >> $OUT 2>&1 || { echo "failed: $*" >> $OUT; exit 1; }; }
N="16384 16384"
#      LB RB W  H seg  grp wpc extra
m $N   8 1 56 4 118  4 8 0
m $N   8 1 56 4 118  4 8 12
m $N   8 1 64 0 128  4 8 0
m $N   8 1 56 4 118  1 8 0
m $N   8 1 56 4 118  4 4 0
m $N   8 1 56 4 118  4 12 0
m $N   8 1 56 4 118  4 16 0
m $N   8 2 56 4 118  4 8 0
m $N   8 4 56 4 118  4 8 0
m $N   8 4 56 4 118  4 4 0
m $N   8 8 56 4 118  4 4 0
m $N   8 2 56 4 118  4 4 0
m $N   16 1 56 4 118  4 8 0
m $N   16 1 56 4 118  4 4 0
m $N   16 2 56 4 118  4 4 0
m $N   16 1 64 0 128  4 4 0
m $N   16 1 64 0 128  1 4 0
m $N   16 1 64 0 128  1 6 0
m $N   8 1 56 4 59   4 8 0
m $N   8 1 56 4 236  4 8 0
m $N   8 1 56 4 472  4 8 0
m $N   8 1 56 4 16384 4 8 0
m 16640 16384  8 1 56 4 118  4 8 0
m 16640 16384  16 1 56 4 118  4 4 0
cat $OUT
