#!/bin/bash
# First GPU pass: whole GPU suite, the bench command as the driver issues it, the contracted arithmetic at
# 6/7/8 steps per launch.  Every GPU step has its own time limit; the chain stops at the first failure.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02_step1
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=10 > $OUT/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc" | tee -a $OUT/gpu_suite.log; tail -5 $OUT/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
for k in 6 7 8; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --arithmetic contracted --max-fuse $k --steps $((k*12)) --warmup $k > $OUT/bench_contracted_k$k.json 2> $OUT/bench_contracted_k$k.err || exit 1
done
timeout -k 10 300 python bench.py --no-cpu-baseline --arithmetic contracted --scheme push --steps 96 --warmup 8 > $OUT/bench_contracted_push.json 2> $OUT/bench_contracted_push.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --arithmetic contracted --precision fp64 --steps 96 --warmup 8 > $OUT/bench_contracted_f64.json 2> $OUT/bench_contracted_f64.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --arithmetic contracted --precision fp64 --max-fuse 6 --steps 96 --warmup 6 > $OUT/bench_contracted_f64_k6.json 2> $OUT/bench_contracted_f64_k6.err || exit 1
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/r02_step1/bench_*.json')):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        r = d['roofline']
        print(p.split('/')[-1], round(d['value']), d['config']['launch_plan'][:4], 'launch_ms %.3f' % r['launch_ms'], 'parity', d.get('parity_checked'), d['parity']['ranks'][0]['bit_identical'], 'traffic', r['traffic'], 'binding', r['binding'], 'frac_bind %.3f' % r['frac_of_binding_roof'])
    except Exception as e:
        print(p, 'unreadable', e)
PY
