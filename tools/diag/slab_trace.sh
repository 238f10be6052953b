#!/bin/bash
# Kernel trace of one rank's 32768 x 4096 slab (RCCL to itself): how long the interior and the edge launches take
cd $GRAFT_REPO_ROOT || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/slab_trace; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o slab -- python3 bench.py --no-cpu-baseline --no-parity --nx 32768 --ny-per-gpu 4096 --slab --steps 40 --warmup 8 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/slab_trace/t/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-40:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8}  {r['Kernel_Name'][:64]}")
PY
