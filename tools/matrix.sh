#!/bin/bash
# Numbers for the documents: the five baseline configurations in both arithmetics, the other layouts
# and spellings, the C++ slab driver over both transports.  Output: gpurun_out/matrix.txt
set -u
OUT=$PWD/gpurun_out/matrix.txt
: > $OUT
say() { echo "$*" | tee -a $OUT; }
cli() { (cd build && timeout -k 10 300 ./main "$@" 2>&1 >/dev/null | tail -n 4 | tr '\n' ' ' | sed 's/time loop only (without allocation and initialisation)/loop/; s/GB.s algorithmic.*peak//') >> $OUT; echo >> $OUT; }
bench() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['config']['workload'], d['config']['arithmetic'], d['config']['layout'], '|', d['config']['decomposition'][:24], '|', round(d['value']), 'MLUPS |', round(d['roofline']['launch_ms'],3), 'ms per launch of', d['roofline']['steps_per_launch'], 'steps')" >> $OUT || exit 1; }
for ar in unrolled contracted; do
  export LBM_ARITHMETIC=$ar
  say "== arithmetic $ar (CLI: LBM_ARITHMETIC=$ar)"
  say "cfg1  ./main -nx 1024 -ny 1024 -w 0.5 -s 50000 -of 500 -omv"
  cli -nx 1024 -ny 1024 -w 0.5 -s 50000 -of 500 -omv
  say "cfg2  ./main -ld -nx 1024 -ny 1024 -u 0.1 -s 10000 ; then -of 1000 -ovf -f file"
  cli -ld -nx 1024 -ny 1024 -u 0.1 -s 10000
  cli -ld -nx 1024 -ny 1024 -u 0.1 -s 10000 -of 1000 -ovf -f /tmp/cfg2_vel.csv
  say "cfg3  ./main -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay x3, then 1000 steps pull / push"
  for i in 1 2 3; do cli -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay; done
  cli -nx 16384 -ny 16384 -s 1000 -of 0 --shear-wave-decay
  cli -nx 16384 -ny 16384 -s 1000 -of 0 --shear-wave-decay -push
  say "cfg3 through -gpus 2 / 4 (slabs on one device): peer copies, then RCCL send/recv"
  for tr in peer rccl; do
    export LBM_SLAB_TRANSPORT=$tr
    cli -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay -gpus 2
    cli -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay -gpus 4
  done
  unset LBM_SLAB_TRANSPORT
  unset LBM_ARITHMETIC
  say "cfg4  lid-driven cavity 16384^2 fp64, pull and push (bench.py)"
  bench --scene lid --precision fp64 --scheme pull --arithmetic $ar
  bench --scene lid --precision fp64 --scheme push --arithmetic $ar
  say "cfg5  32768^2 on one GPU; 32768 x 4096 slab through RCCL to self"
  bench --nx 32768 --ny-global 32768 --steps 40 --warmup 8 --arithmetic $ar
  bench --nx 32768 --ny-per-gpu 4096 --slab --arithmetic $ar
  say "other flavours at 16384^2"
  bench --scheme push --arithmetic $ar
  bench --scene lid --scheme push --arithmetic $ar
  bench --scene lid --scheme pull --arithmetic $ar
  bench --precision fp64 --arithmetic $ar
  bench --precision fp64 --scheme push --arithmetic $ar
  bench --max-fuse 1 --arithmetic $ar
  say "smaller lattices (fp32 pull; 1024^2 also cavity push and fp64)"
  for n in 512 1024 1536 2048 4096; do bench --nx $n --ny-global $n --steps 1000 --warmup 100 --arithmetic $ar; done
  bench --nx 1024 --ny-global 1024 --steps 1000 --warmup 100 --scene lid --scheme push --arithmetic $ar
  bench --nx 1024 --ny-global 1024 --steps 1000 --warmup 100 --precision fp64 --arithmetic $ar
done
say "== [NY][NX][9] arrays and the rolled arithmetic (strict rounding)"
bench --layout aos
bench --layout aos --precision fp64
bench --layout aos --scene lid --scheme push
say "== [NY][NX][9] arrays, contracted arithmetic"
bench --layout aos --arithmetic contracted
bench --layout aos --precision fp64 --arithmetic contracted
bench --layout aos --scene lid --scheme push --arithmetic contracted
bench --rolled
bench --rolled --scheme push
say "== odd row lengths"
bench --nx 22627 --ny-per-gpu 22627 --precision fp64 --arithmetic contracted
bench --nx 16383 --ny-per-gpu 16384 --arithmetic unrolled
say "== the device while loaded (clock and power of the box this ran on)"
(timeout -k 5 60 python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-parity > /dev/null 2>&1 &
 sleep 12; timeout 10 rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | tr -s ' ' >> $OUT; wait)
