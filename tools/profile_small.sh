#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_small
mkdir -p $OUT
cd build
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d ../$OUT/pull -- ./main -nx 1024 -ny 1024 -of 0 -s 2000 > ../$OUT/pull.log 2>&1; echo "exit $?"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d ../$OUT/lid -- ./main -nx 1024 -ny 1024 -of 0 -s 2000 -ld > ../$OUT/lid.log 2>&1; echo "exit $?"
cd ..
for k in pull lid; do cat $OUT/$k/*/*_kernel_stats.csv | cut -c1-160 | head -4; tail -1 $OUT/$k.log; python3 - $OUT/$k <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'vec' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
gaps=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']) for i in range(len(rows)-1)]
durs=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
import statistics as st
print("n",len(rows),"median dur ns",st.median(durs),"median gap ns",st.median(gaps))
PY
done
