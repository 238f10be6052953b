#!/bin/bash
# cavity sweeps, contracted fp32: seven against eight steps per launch
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
for rep in 1 2; do
for k in 7 8; do
  one --scene lid --max-fuse $k --steps 112 2>/dev/null
  one --scene lid --scheme push --max-fuse $k --steps 112 2>/dev/null
done
done
