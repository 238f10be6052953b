#!/bin/bash
OUT=gpurun_out/small_seg.txt
: > $OUT
for k in 4 6; do
  for seg in 0 4 5 6 8 10 12 16 24; do
    steps=$((k*200))
    r=$(LBM_FUSED_SEG_ROWS=$seg timeout -k 5 120 python3 bench.py --no-cpu-baseline --no-parity --nx 1024 --ny-per-gpu 1024 --steps $steps --warmup $((k*10)) --max-fuse $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['launch_ms']*1000,1))") || exit 1
    echo "1024^2 K=$k seg=$seg: $r (MLUPS, us per launch)" | tee -a $OUT
  done
done
