#!/bin/bash
OUT=gpurun_out/step4.txt
: > $OUT
b() { tag="$1"; shift; r=$(timeout -k 5 200 python3 bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config']['launch_plan'][:1], round(d['roofline']['launch_ms']*1000,1))") || exit 1; echo "$tag: $r" | tee -a $OUT; }
b "lid push 16384" --scene lid --scheme push
b "lid pull 16384" --scene lid --scheme pull
b "lid push strict 16384" --scene lid --scheme push --arithmetic unrolled
b "lid push 1024" --scene lid --scheme push --nx 1024 --ny-per-gpu 1024 --steps 1000
b "lid pull 1024" --scene lid --scheme pull --nx 1024 --ny-per-gpu 1024 --steps 1000
b "lid push 2048" --scene lid --scheme push --nx 2048 --ny-per-gpu 2048 --steps 400
b "shear 1024" --nx 1024 --ny-per-gpu 1024 --steps 1000
b "shear 512" --nx 512 --ny-per-gpu 512 --steps 1000
b "shear 16384" 
LBM_FUSED_XCD=1 b "shear 16384 xcd map"
LBM_FUSED_WALL_SPLIT=0 b "lid push 1024 no wall split" --scene lid --scheme push --nx 1024 --ny-per-gpu 1024 --steps 1000
