#!/bin/bash
# PMC passes for one variant of tools/tune_fused: $1 = variant filter, $2 = out tag
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$2
mkdir -p $OUT
B="$GRAFT_REPO_ROOT/tools/tune_fused 16384 16384 f32 2 5 $1"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1 || exit 1
echo done $2
