#!/bin/bash
OUT=gpurun_out/small_sweep.txt
: > $OUT
for n in 512 1024 2048; do
  for k in 1 2 3 4 5 6 8; do
    steps=$((k*200))
    r=$(timeout -k 5 120 python3 bench.py --no-cpu-baseline --no-parity --nx $n --ny-per-gpu $n --steps $steps --warmup $((k*10)) --max-fuse $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config']['launch_plan'][:2], round(d['roofline']['launch_ms']*1000,1))") || exit 1
    echo "$n^2 max-fuse $k: $r (MLUPS, plan, us per launch)" | tee -a $OUT
  done
done
