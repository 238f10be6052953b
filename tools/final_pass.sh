#!/bin/bash
# Last GPU pass of a round: the whole GPU suite, smoke(), the bench command as the driver issues it and with
# its defaults, a two-rank rehearsal of the N>1 line, then the rocprofv3 passes of both bench commands.
# Every GPU step has its own time limit; the chain stops at the first failure.  usage: final_pass.sh TAG
set -u
export TMPDIR=/tmp
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread --durations=8 > $OUT/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc" | tee -a $OUT/gpu_suite.log; tail -12 $OUT/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
LBM_BENCH_REHEARSE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_rehearsal_2_ranks_gloo_one_gpu.json 2> $OUT/bench_rehearsal.err || exit 1
python - "$OUT" <<'PY'
import json, glob, sys
for p in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d['roofline']
    print(p.split('/')[-1], round(d['value']), d['arithmetic'], '|', d['config']['workload'], d['config']['launch_plan'][:4],
          'launch_ms %.3f' % r['launch_ms'], 'parity', d.get('parity_checked'), 'bound', r['bound'], 'frac %.3f' % r['frac'],
          'strict', round(d['strict_arithmetic']['value']) if d.get('strict_arithmetic') else None)
PY
bash tools/profile_bench.sh ${TAG}_20_5 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/profile_20_5.log 2>&1 || { tail -5 $OUT/profile_20_5.log; exit 1; }
tail -3 $OUT/profile_20_5.log
bash tools/profile_bench.sh ${TAG}_100_10 --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/profile_100_10.log 2>&1 || { tail -5 $OUT/profile_100_10.log; exit 1; }
tail -3 $OUT/profile_100_10.log
