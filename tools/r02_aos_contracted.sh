#!/bin/bash
# GPU pass for the contracted arithmetic on [NY][NX][9] arrays: the whole GPU suite, the bench in that layout
# in both arithmetics and precisions, and a rehearsal of the multi-rank bench (two gloo ranks on the one GPU).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02_aos_contracted
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread --durations=5 > $OUT/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc" | tee -a $OUT/gpu_suite.log; tail -9 $OUT/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
for prec in fp32 fp64; do
  for arith in unrolled contracted; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --layout aos --arithmetic $arith --precision $prec > $OUT/bench_aos_${arith}_$prec.json 2> $OUT/bench_aos_${arith}_$prec.err || { tail -5 $OUT/bench_aos_${arith}_$prec.err; exit 1; }
  done
done
timeout -k 10 300 python bench.py --no-cpu-baseline --layout aos --arithmetic contracted --scene lid --scheme push > $OUT/bench_aos_contracted_lid_push.json 2> $OUT/bench_aos_contracted_lid_push.err || exit 1
GPU_MAX_HW_QUEUES=2 LBM_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_rehearsal_2_ranks.json 2> $OUT/bench_rehearsal_2_ranks.err || { tail -5 $OUT/bench_rehearsal_2_ranks.err; exit 1; }
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/r02_aos_contracted/bench_*.json')):
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d['roofline']
    print(p.split('/')[-1], round(d['value']), d['config'].get('launch_plan', [])[:4], 'launch_ms %.3f' % r['launch_ms'],
          'parity', d.get('parity_checked'), [x.get('bit_identical') for x in d.get('parity', {}).get('ranks', [])], 'n_gpus', d['n_gpus'])
PY
