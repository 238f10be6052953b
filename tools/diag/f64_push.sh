#!/bin/bash
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], d['config']['layout'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3], d['parity_checked'])"; }
one --precision fp64 --scheme push
one --precision fp64 --scheme push --arithmetic unrolled
one --precision fp64 --scheme push --scene lid --no-parity
one --precision fp64 --scheme push --scene lid --arithmetic unrolled --no-parity
one --precision fp64
