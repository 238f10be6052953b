#!/bin/bash
# 16384^2, 100 steps: where the command-line driver's time loop goes, launch by launch (kernel trace)
cd $GRAFT_REPO_ROOT || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/cli_vs_bench; rm -rf $OUT; mkdir -p $OUT
for s in 100 200 400; do (cd build && timeout -k 10 120 ./main -nx 16384 -ny 16384 -s $s -of 0 --shear-wave-decay 2>&1 | grep "time loop" | cut -c1-100) >> $OUT/cli.txt || exit 1; done
(cd build && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_cli -o cli -- ./main -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay > $OUT/prof_cli.log 2>&1) || exit 1
cat $OUT/cli.txt
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ.get('OUT', 'gpurun_out/cli_vs_bench') + '/prof_cli/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows[:24]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e6:9.3f} ms  +gap {(s - prev_end) / 1e6:7.3f}  dur {(e - s) / 1e6:7.3f}  {r['Kernel_Name'][:70]}")
    prev_end = e
PY
