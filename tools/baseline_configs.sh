#!/bin/bash
# The five configurations named as baselines, on one MI355X.  Output: gpurun_out/baseline_configs.txt
set -u
OUT=$PWD/gpurun_out/baseline_configs.txt
: > $OUT
say() { echo "$*" >> $OUT; }
cli() { (cd build && timeout -k 10 300 ./main "$@" 2>&1 >/dev/null | tail -n 4 | tr '\n' ' ') >> $OUT; echo >> $OUT; }
bench() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['config']['workload'], '|', d['config']['decomposition'][:28], '|', round(d['value']), 'MLUPS |', round(d['roofline']['launch_ms'],3), 'ms per launch of', round(d['roofline']['steps_per_launch'],2), 'steps')" >> $OUT; }
say "cfg1  ./main -nx 1024 -ny 1024 -w 0.5 -s 50000 -of 500 -omv (fp32 pull; last two stderr lines: time loop only, whole run)"
cli -nx 1024 -ny 1024 -w 0.5 -s 50000 -of 500 -omv
say "cfg2  ./main -ld -nx 1024 -ny 1024 -u 0.1 -s 10000 (fp32, push as the reference; then -ovf every 1000 into a file)"
cli -ld -nx 1024 -ny 1024 -u 0.1 -s 10000
cli -ld -nx 1024 -ny 1024 -u 0.1 -s 10000 -of 1000 -ovf -f /tmp/cfg2_vel.csv
say "cfg3  ./main -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay, five repeats (fp32 pull), then 1000 steps"
for i in 1 2 3 4 5; do cli -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay; done
cli -nx 16384 -ny 16384 -s 1000 -of 0 --shear-wave-decay
cli -nx 16384 -ny 16384 -s 1000 -of 0 --shear-wave-decay -push
say "cfg4  lid-driven cavity 16384^2 fp64, pull and push (bench.py)"
bench --scene lid --precision fp64 --scheme pull
bench --scene lid --precision fp64 --scheme push
say "cfg5  shear wave fp32 pull 32768^2 on one GPU (the P=1 point of strong scaling) and a 32768 x 4096 slab through RCCL (the P=1 point of weak scaling)"
bench --nx 32768 --ny-global 32768 --steps 40 --warmup 8
bench --nx 32768 --ny-per-gpu 4096 --slab
cat $OUT
