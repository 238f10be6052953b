#!/bin/bash
# Engine clock of each launch of a command-line run (GRBM_GUI_ACTIVE cycles / launch duration): is the slow
# start of a process a clock ramp?
cd $GRAFT_REPO_ROOT || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/cli_ramp; rm -rf $OUT; mkdir -p $OUT
(cd build && timeout -s KILL 100 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/t -o cli -- ./main -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay > $OUT/log.txt 2>&1) || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/cli_ramp/t/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_fused' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    c = float(r['Counter_Value'])
    print(f"dur {d / 1e6:6.3f} ms  cycles {c / 1e6:8.2f} M  clock {c / d:5.2f} GHz")
PY
