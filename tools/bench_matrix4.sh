#!/bin/bash
set -u
out=${1:-gpurun_out/bench_matrix4.log}
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
for s in pull push; do
for cfg in "128 4" "128 6" "128 8" "256 4" "256 8" "512 8"; do set -- $cfg
run "fp32 $s 16384^2 tx=$1 wpc=$2" --steps 30 --warmup 3 --scheme $s --threads-x $1 --waves-per-cu $2; done; done
for s in pull push; do
for cfg in "128 4" "128 6" "256 4" "256 8"; do set -- $cfg
run "fp64 $s 16384^2 tx=$1 wpc=$2" --steps 20 --warmup 3 --precision fp64 --scheme $s --threads-x $1 --waves-per-cu $2; done; done
for n in 8192 4096 2048; do for w in 5 6 8 12; do run "fp32 pull ${n}^2 wpc=$w" --steps 200 --warmup 10 --nx $n --ny-per-gpu $n --waves-per-cu $w; done; done
for w in 5 6; do run "fp32 pull 32768x8192 wpc=$w" --steps 30 --warmup 3 --nx 32768 --ny-per-gpu 8192 --waves-per-cu $w; done
for w in 5 6; do run "fp32 pull 1000x1000 wpc=$w" --steps 2000 --warmup 10 --nx 1000 --ny-per-gpu 1000 --waves-per-cu $w; done
cat "$out"
