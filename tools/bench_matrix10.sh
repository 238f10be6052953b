#!/bin/bash
set -u
out=${1:-gpurun_out/bench_matrix10.log}
: > "$out"
run() {
  label="$1"; shift
  line=$(timeout -k 10 300 python -u bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1)
  python - "$label" "$line" >> "$out" <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print(f"{label:44s} {d['value']:10.0f} MLUPS  {d['roofline']['achieved']:8.1f} GB/s  frac {d['roofline']['frac']:.3f}")
except Exception as e:
    print(f"{label:44s} FAILED {e} {line[:200]}")
PY
}
for x in 0 2; do export LBM_XCD_ROWS=$x
for w in 5 6 7 8; do run "xcd=$x fp32 pull wpc=$w" --steps 30 --warmup 3 --waves-per-cu $w; done
for w in 6 7; do run "xcd=$x fp32 push wpc=$w" --steps 30 --warmup 3 --scheme push --waves-per-cu $w; done
for w in 4 5 6; do run "xcd=$x fp64 pull wpc=$w" --steps 20 --warmup 3 --precision fp64 --waves-per-cu $w; done
done
unset LBM_XCD_ROWS
run "fp32 pull 1024^2 default" --steps 3000 --warmup 100 --nx 1024 --ny-per-gpu 1024
run "fp32 lid push 1024^2 default" --steps 3000 --warmup 100 --nx 1024 --ny-per-gpu 1024 --scene lid --scheme push
cat "$out"
