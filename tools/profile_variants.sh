#!/bin/bash
# PMC passes for other sweep flavours (GPU box).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_variants
mkdir -p $OUT
for v in "push fp32" "pull fp64" "push fp64"; do set -- $v
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $OUT/${1}_${2}_$c -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --scheme $1 --precision $2 > $OUT/${1}_${2}_$c.log 2>&1; echo "$1 $2 $c exit $?"
  done
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/prof_variants/*_SIZE")):
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        rows=[r for r in csv.DictReader(open(f)) if 'periodic_vec' in r['Kernel_Name']]
        v=[float(r['Counter_Value']) for r in rows]
        print(d.split('/')[-1], len(v), sum(v)/len(v))
PY
