#!/bin/bash
# A/B on one box: the library in the tree, then build/before/liblbm_hip.so copied over it (the box's copy only)
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
suite() {
  one --precision fp64
  one --precision fp64 --scheme push
  one --precision fp64 --scene lid
  one --precision fp64 --scene lid --scheme push
  one --precision fp64 --arithmetic unrolled
  for n in 1024 2048 4096; do
    one --precision fp64 --nx $n --ny-global $n --steps 700 --warmup 70
  done
}
echo "== tree"; suite 2>/dev/null
cp build/before/liblbm_hip.so lbm_amd/liblbm_hip.so
echo "== before"; suite 2>/dev/null
