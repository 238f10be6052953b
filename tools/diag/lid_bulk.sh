#!/bin/bash
# cavity sweeps, every flavour, large and small
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
for a in contracted unrolled; do
  one --scene lid --arithmetic $a
  one --scene lid --scheme push --arithmetic $a
  one --scene lid --precision fp64 --arithmetic $a
  one --scene lid --scheme push --precision fp64 --arithmetic $a
  one --scene lid --scheme push --nx 1024 --ny-global 1024 --steps 1000 --warmup 100 --arithmetic $a
  one --scene lid --scheme pull --nx 1024 --ny-global 1024 --steps 1000 --warmup 100 --arithmetic $a
  one --scene lid --scheme push --nx 2048 --ny-global 2048 --steps 1000 --warmup 100 --arithmetic $a
done
one --scene lid --max-fuse 8 --steps 96
one --scene lid --scheme push --max-fuse 8 --steps 96
