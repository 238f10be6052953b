#!/bin/bash
# Counters of the marching access pattern (tools/micro/march) at several segment lengths: address translation,
# read latency and outstanding reads at the L2's memory side.  One pass per counter group and setting.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_march
rm -rf $OUT; mkdir -p $OUT
B=$PWD/tools/micro/march
run() {  # tag, counters..., then -- march args
  tag=$1; shift
  ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -s KILL 120 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/$tag -- $B "$@" > $OUT/$tag.log 2>&1 || { echo "$tag failed"; tail -3 $OUT/$tag.log; exit 1; }
}
for S in 1 16 118; do
  A="16384 16384 8 1 56 4 $S 4 8 0 0 0"
  run tlb_S$S TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum -- $A
  run ea_S$S TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum -- $A
  run ea2_S$S TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum -- $A
  run tcp2_S$S TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum -- $A
  grep march $OUT/tlb_S$S.log
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*/")):
    for p in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if "k_march" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d.rstrip("/").split("/")[-1], {k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()})
PY
