#!/bin/bash
# Last GPU pass of the round: the whole GPU suite, smoke(), the bench command as the driver issues it and with
# its defaults, then the rocprofv3 passes of the driver's command.  Every GPU step has its own time limit; the
# chain stops at the first failure.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02_final
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread --durations=8 > $OUT/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc" | tee -a $OUT/gpu_suite.log; tail -12 $OUT/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/r02_final/bench_*.json')):
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d['roofline']
    print(p.split('/')[-1], round(d['value']), d['config']['launch_plan'][:4], 'launch_ms %.3f' % r['launch_ms'],
          'parity', d.get('parity_checked'), 'traffic', r['traffic'], 'binding', r['binding'],
          'frac_bind %.3f' % r['frac_of_binding_roof'], 'strict', round(d['strict_arithmetic']['value']))
PY
bash tools/r02_profile_bench.sh final_20_5 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/profile_20_5.log 2>&1 || { tail -5 $OUT/profile_20_5.log; exit 1; }
tail -3 $OUT/profile_20_5.log
