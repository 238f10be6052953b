#!/bin/bash
# fp64 sweeps with the library in the tree, large and small lattices, both arithmetics (bench.py --no-cpu-baseline)
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
for a in contracted unrolled; do
  one --precision fp64 --arithmetic $a
  one --precision fp64 --arithmetic $a --scheme push
  one --precision fp64 --arithmetic $a --scene lid
  one --precision fp64 --arithmetic $a --scene lid --scheme push
done
for n in 1024 2048 4096; do
  one --precision fp64 --nx $n --ny-global $n --steps 700 --warmup 70
  one --precision fp64 --nx $n --ny-global $n --steps 700 --warmup 70 --scene lid --scheme push
done
