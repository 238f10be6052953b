#!/bin/bash
# HIP API calls of the command-line driver between its clock start and the first sweep (where the set-up time goes)
cd $GRAFT_REPO_ROOT || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/cli_api; rm -rf $OUT; mkdir -p $OUT
(cd build && timeout -k 10 200 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $OUT/t -o cli -- ./main -nx 16384 -ny 16384 -s 100 -of 0 --shear-wave-decay > $OUT/log.txt 2>&1) || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob
api = glob.glob('gpurun_out/cli_api/t/**/*hip_api_trace.csv', recursive=True)[0]
ker = glob.glob('gpurun_out/cli_api/t/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function']) for r in csv.DictReader(open(api))]
rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'KERNEL ' + r['Kernel_Name'][:50]) for r in csv.DictReader(open(ker))]
rows.sort()
# from the two big hipMalloc calls on
first = next(i for i, r in enumerate(rows) if 'k_reciprocal' in r[2])
t0 = rows[first][0]
for s, e, f in rows[first:first + 60]:
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  {f}")
PY
