#!/bin/bash
set -u
out=${1:-gpurun_out/bench_matrix11.log}
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
run "fp64 pull 22627^2 (odd)" --steps 20 --warmup 3 --precision fp64 --nx 22627 --ny-per-gpu 22627
run "fp64 push 22627^2 (odd)" --steps 20 --warmup 3 --precision fp64 --nx 22627 --ny-per-gpu 22627 --scheme push
run "fp64 pull 22628^2 (even)" --steps 20 --warmup 3 --precision fp64 --nx 22628 --ny-per-gpu 22628
run "fp32 pull 32000^2" --steps 20 --warmup 3 --nx 32000 --ny-per-gpu 32000
run "fp32 push 32000^2" --steps 20 --warmup 3 --nx 32000 --ny-per-gpu 32000 --scheme push
run "fp32 pull 10002^2 (nx%4=2)" --steps 50 --warmup 3 --nx 10002 --ny-per-gpu 10002
run "fp32 pull 10000^2" --steps 50 --warmup 3 --nx 10000 --ny-per-gpu 10000
run "fp32 pull 3162^2 (nx%4=2)" --steps 300 --warmup 10 --nx 3162 --ny-per-gpu 3162
run "fp32 pull 3164^2" --steps 300 --warmup 10 --nx 3164 --ny-per-gpu 3164
cat "$out"
