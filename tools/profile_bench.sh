#!/bin/bash
# rocprofv3 passes of the driver's own bench command (GPU box); each pass is its own run.
# usage: profile_bench.sh TAG [bench.py arguments...]
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 bench.py $*"
echo "$BENCH" > $OUT/command.txt
timeout -k 10 300 $BENCH > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_clk -- $BENCH > $OUT/pmc_clk.log 2>&1 || echo "clock pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
summary = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_clk"):
    for p in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "k_fused" not in k and "k_pull" not in k and "k_push" not in k: continue
            for c, v in cs.items():
                summary.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
for p in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Name"].split("(")[0]
        if k in summary:
            summary[k]["kernel_trace"] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"])}
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
