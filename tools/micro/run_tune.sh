#!/bin/bash
# runs every tools/micro/tune_k_* binary given as arguments, each under its own time limit
OUT=gpurun_out/tune_k.txt
: > $OUT
for t in "$@"; do
  timeout -k 5 90 tools/micro/tune_k_$t $TK_ARGS >> $OUT 2>&1 || { echo "$t failed rc=$?" >> $OUT; exit 1; }
done
grep -v "^check.*mismatches 0$" $OUT
