#!/bin/bash
# GPU pass: whole GPU suite, then the bench matrix in both arithmetics.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02_step5
mkdir -p $OUT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=8 > $OUT/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc" | tee -a $OUT/gpu_suite.log; tail -4 $OUT/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
b() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { echo "bench $tag failed"; tail -5 $OUT/bench_$tag.err; exit 1; }; }
b strict_20_5 --steps 20 --warmup 5 --arithmetic unrolled
b contracted_20_5 --steps 20 --warmup 5 --arithmetic contracted
b strict --arithmetic unrolled
b contracted --arithmetic contracted
b contracted_push --arithmetic contracted --scheme push
b contracted_f64 --arithmetic contracted --precision fp64
b contracted_f64_push --arithmetic contracted --precision fp64 --scheme push
b contracted_lid_push --arithmetic contracted --scene lid --scheme push
b contracted_lid_pull --arithmetic contracted --scene lid --scheme pull
b strict_lid_push --scene lid --scheme push --arithmetic unrolled
b strict_f64 --precision fp64 --arithmetic unrolled
b strict_k1 --max-fuse 1 --arithmetic unrolled
b contracted_slab --arithmetic contracted --slab
b contracted_32768 --arithmetic contracted --nx 32768 --ny-per-gpu 32768 --steps 40
b contracted_1024 --arithmetic contracted --nx 1024 --ny-per-gpu 1024 --steps 2000 --warmup 100
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/r02_step5/bench_*.json')):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        r = d['roofline']
        print('%-34s %8.0f plan %s launch_ms %.3f parity %s/%s' % (p.split('/')[-1], d['value'], d['config']['launch_plan'][:3], r['launch_ms'], d.get('parity_checked'), d['parity']['ranks'][0]['bit_identical']))
    except Exception as e:
        print(p, 'unreadable', e)
PY
