#!/bin/bash
# small lattices, five steps per launch: halos of whole lanes (default) against whole 32-byte granules
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
for n in 512 768 1024; do
  for narrow in 1 0; do
    echo "== ${n}^2 LBM_FUSED_NARROW=$narrow"
    export LBM_FUSED_NARROW=$narrow
    one --nx $n --ny-global $n --steps 1000 --warmup 100
    one --nx $n --ny-global $n --steps 1000 --warmup 100 --arithmetic unrolled
    one --nx $n --ny-global $n --steps 1000 --warmup 100 --scene lid --scheme push
    one --nx $n --ny-global $n --steps 1000 --warmup 100 --precision fp64
  done
done
