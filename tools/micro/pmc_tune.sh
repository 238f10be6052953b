#!/bin/bash
# counters of one tune_k binary ($1), separate passes
export TMPDIR=/tmp
B=tools/micro/tune_k_$1
OUT=gpurun_out/pmc_tune_$1
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq1", "sq2", "fetch", "write"):
    for p in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "k_fused" not in k: continue
            print(sub, k)
            for c, v in cs.items():
                print(f"    {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
for p in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    print(open(p).read())
PY
