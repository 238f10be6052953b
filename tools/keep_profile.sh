#!/bin/bash
# keep_profile.sh gpurun_out/prof_TAG profiles/rNN/NAME: the files of one tools/profile_bench.sh run that are worth keeping
set -eu
SRC=$1; DST=$2
mkdir -p $DST
cp $SRC/bench.json $SRC/summary.json $SRC/command.txt $DST/
for sub in pmc_fetch pmc_write pmc_sq pmc_clk; do
  f=$(find $SRC/$sub -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $DST/${sub}_counter_collection.csv
done
f=$(find $SRC/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $DST/trace_kernel_stats.csv
ls $DST
