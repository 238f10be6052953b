#!/bin/bash
set -u
out=${1:-gpurun_out/bench_matrix6.log}
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
export LBM_HALF_VEC=2
for w in 2 3 4; do run "double fp64 pull wpc=$w" --steps 20 --warmup 3 --precision fp64 --waves-per-cu $w; done
for w in 2 3 4; do run "double fp64 push wpc=$w" --steps 20 --warmup 3 --precision fp64 --scheme push --waves-per-cu $w; done
for w in 2 3 4; do run "double fp32 pull wpc=$w" --steps 20 --warmup 3 --waves-per-cu $w; done
for w in 2 3 4; do run "double fp32 push wpc=$w" --steps 20 --warmup 3 --scheme push --waves-per-cu $w; done
cat "$out"
