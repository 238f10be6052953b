#!/bin/bash
# 1024^2 fp32, five steps per launch: the cavity launch against the periodic one (kernel trace: duration, period, grid)
export TMPDIR=/tmp
OUT=gpurun_out/small_cav; rm -rf $OUT; mkdir -p $OUT
for v in "shear push" "lid push" "shear pull" "lid pull"; do
  set -- $v
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$1_$2 -- python3 bench.py --no-cpu-baseline --no-parity --scene $1 --scheme $2 --nx 1024 --ny-per-gpu 1024 --steps 400 --warmup 40 > $OUT/$1_$2.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob
for d in sorted(glob.glob('gpurun_out/small_cav/*_*/')):
    p = glob.glob(d + '**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(p)) if 'k_fused' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[20:120]
    dur = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    per = sorted(int(b['Start_Timestamp']) - int(a['Start_Timestamp']) for a, b in zip(rows, rows[1:]))
    r0 = rows[0]
    print(d.split('/')[-2], r0['Kernel_Name'][:60], 'grid', r0.get('Grid_Size_X'), 'wg', r0.get('Workgroup_Size_X'), 'lds', r0.get('LDS_Block_Size'), 'vgpr', r0.get('VGPR_Count'),
          'dur median', dur[len(dur)//2], 'min', dur[0], 'period median', per[len(per)//2])
PY
