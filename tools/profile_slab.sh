#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_slab
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --slab > $OUT/trace.log 2>&1; echo "exit $?"
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-150 | head -12
python3 - $OUT/trace <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*_kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=[r for r in rows if 'k_pull_periodic_vec' in r['Kernel_Name'] and int(r['End_Timestamp'])-int(r['Start_Timestamp'])>1_000_000]
t0=int(big[5]['Start_Timestamp'])
for r in rows:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    if 0<=s<7_000_000:
        print(f"{s/1e3:10.1f} {e/1e3:10.1f} us  dur {(e-s)/1e3:8.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:70]}  grid {r['Grid_Size']}")
PY
