#!/bin/bash
set -e
one() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['config']['arithmetic'], '|', round(d['value']), 'MLUPS |', d['config']['launch_plan'][:3])"; }
cp lbm_amd/liblbm_hip.so /tmp/tree_liblbm_hip.so
for rep in 1 2; do
echo "== tree"; one --rolled --scheme push; one --rolled
cp build/before/liblbm_hip.so lbm_amd/liblbm_hip.so
echo "== before"; one --rolled --scheme push; one --rolled
cp /tmp/tree_liblbm_hip.so lbm_amd/liblbm_hip.so
done
