#!/bin/bash
set -u
out=${1:-gpurun_out/bench_matrix7.log}
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
for w in 4 5 6 7 8; do run "fp32 push wpc=$w" --steps 30 --warmup 3 --scheme push --waves-per-cu $w; done
for w in 4 5 6 7; do run "fp64 push wpc=$w" --steps 20 --warmup 3 --precision fp64 --scheme push --waves-per-cu $w; done
cat "$out"
