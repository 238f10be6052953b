#!/bin/bash
# Segment length against throughput on large lattices.  Output: gpurun_out/seg_sweep.txt
set -u
OUT=gpurun_out/seg_sweep.txt
: > $OUT
for cfg in "--nx 16384 --ny-per-gpu 16384" "--nx 16384 --ny-per-gpu 16384 --precision fp64" "--nx 8192 --ny-per-gpu 8192 --steps 200 --warmup 20" "--nx 32768 --ny-per-gpu 32768 --steps 40 --warmup 8" "--nx 16384 --ny-per-gpu 16384 --scene lid --scheme push" "--nx 12288 --ny-per-gpu 12288"; do
  echo "== $cfg" >> $OUT
  for s in 0 112 128 144 160 171 176 192; do
    LBM_FUSED_SEG_ROWS=$s timeout -k 10 200 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  seg $s', round(d['value']), d['roofline']['launch_ms'])" >> $OUT || exit 1
  done
done
cat $OUT
