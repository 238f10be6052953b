#!/bin/bash
# A/B on one box: the library in the tree, then build/before/liblbm_hip.so copied over it (the box's
# copy only), then the tree's again.
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
suite() {
  one
  one --scheme push
  one --scene lid
  one --scene lid --scheme push
  one --arithmetic unrolled
  one --precision fp64
  one --precision fp64 --scheme push
  one --precision fp64 --scene lid
  one --precision fp64 --scene lid --scheme push
  one --precision fp64 --arithmetic unrolled
  for n in 1024 2048 4096; do
    one --nx $n --ny-global $n --steps 700 --warmup 70
    one --precision fp64 --nx $n --ny-global $n --steps 700 --warmup 70
  done
}
cp lbm_amd/liblbm_hip.so /tmp/tree_liblbm_hip.so
echo "== tree"; suite 2>/dev/null
cp build/before/liblbm_hip.so lbm_amd/liblbm_hip.so
echo "== before"; suite 2>/dev/null
cp /tmp/tree_liblbm_hip.so lbm_amd/liblbm_hip.so
echo "== tree again"; suite 2>/dev/null
