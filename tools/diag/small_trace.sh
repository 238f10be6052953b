#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/small_trace
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --no-cpu-baseline --no-parity --nx 1024 --ny-per-gpu 1024 --steps 400 --warmup 40 > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
p = glob.glob('gpurun_out/small_trace/t/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(p)) if 'k_fused' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[20:120]
dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
per = [int(b['Start_Timestamp']) - int(a['Start_Timestamp']) for a, b in zip(rows, rows[1:])]
gap = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows, rows[1:])]
print('kernel', rows[0]['Kernel_Name'][:80], 'grid', rows[0].get('Grid_Size_X'), 'wg', rows[0].get('Workgroup_Size_X'))
print('duration ns: median', sorted(dur)[len(dur)//2], 'min', min(dur), 'max', max(dur))
print('period   ns: median', sorted(per)[len(per)//2])
print('gap      ns: median', sorted(gap)[len(gap)//2], 'min', min(gap), 'max', max(gap))
PY
