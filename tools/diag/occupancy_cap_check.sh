#!/bin/bash
# flavours whose multi-step kernels use few registers and rings in LDS (rolled, [NY][NX][9], odd rows, fp64 push)
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], d['config']['layout'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
one --rolled
one --rolled --scheme push
one --layout aos
one --layout aos --precision fp64
one --layout aos --scene lid --scheme push
one --nx 16383 --ny-per-gpu 16384 --arithmetic unrolled
one --nx 22627 --ny-per-gpu 22627 --precision fp64 --arithmetic contracted
one --precision fp64 --scheme push
one --precision fp64 --scheme push --arithmetic unrolled
one --scheme push --arithmetic unrolled
one
