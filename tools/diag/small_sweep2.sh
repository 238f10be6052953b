#!/bin/bash
OUT=gpurun_out/small_sweep2.txt
: > $OUT
for spec in "512 512" "1024 1024" "1536 1536" "2048 2048" "3072 3072" "4096 4096"; do
  set -- $spec; n=$1
  for k in 3 4 5 6 7 8; do
    steps=$((k*100))
    r=$(timeout -k 5 120 python3 bench.py --no-cpu-baseline --no-parity --nx $n --ny-per-gpu $n --steps $steps --warmup $((k*10)) --max-fuse $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config']['launch_plan'][:1], round(d['roofline']['launch_ms']*1000,1))") || exit 1
    echo "$n^2 max-fuse $k: $r" | tee -a $OUT
  done
done
for k in 4 6; do r=$(timeout -k 5 120 python3 bench.py --no-cpu-baseline --no-parity --scene lid --scheme push --nx 1024 --ny-per-gpu 1024 --steps $((k*100)) --warmup $((k*10)) --max-fuse $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config']['launch_plan'][:1], round(d['roofline']['launch_ms']*1000,1))") || exit 1; echo "lid push 1024^2 max-fuse $k: $r" | tee -a $OUT; done
