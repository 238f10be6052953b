#!/bin/bash
# small cavities with (default) and without the periodic code in the bulk
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
for rep in 1 2; do
for b in 1 0; do
  echo "== LBM_FUSED_BULK=$b"
  export LBM_FUSED_BULK=$b
  for n in 512 1024 2048 4096; do
    one --scene lid --scheme push --nx $n --ny-global $n --steps 1000 --warmup 100
  done
  one --scene lid --scheme pull --nx 1024 --ny-global 1024 --steps 1000 --warmup 100
  one --scene lid --scheme push --nx 1024 --ny-global 1024 --steps 1000 --warmup 100 --arithmetic unrolled
done
done
