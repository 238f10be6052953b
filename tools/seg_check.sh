#!/bin/bash
# Default segment choice against a fixed 128 rows on large lattices.  Output: gpurun_out/seg_check.txt
set -u
OUT=gpurun_out/seg_check.txt
: > $OUT
for cfg in "--nx 16384 --ny-per-gpu 16384" "--nx 16384 --ny-per-gpu 16384 --scheme push" "--nx 16384 --ny-per-gpu 16384 --precision fp64" "--nx 16384 --ny-per-gpu 16384 --scene lid --scheme pull --precision fp64" "--nx 32768 --ny-per-gpu 32768 --steps 40 --warmup 8" "--nx 22627 --ny-per-gpu 22627 --precision fp64 --steps 40 --warmup 8" "--nx 20000 --ny-per-gpu 20000 --steps 60 --warmup 12" "--nx 16384 --ny-per-gpu 16384 --layout aos" "--slab" "--nx 16384 --ny-per-gpu 16384 --rolled"; do
  echo "== $cfg" >> $OUT
  for s in 128 0 128 0; do
    LBM_FUSED_SEG_ROWS=$s timeout -k 10 200 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  seg $s', round(d['value']), round(d['roofline']['launch_ms'],3))" >> $OUT || exit 1
  done
done
cat $OUT
