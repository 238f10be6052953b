#!/bin/bash
# build/main_slabtest: -gpus N against one slab, text for text
cd build
M=./main_slabtest
fail=0
cmp_run() { # args...
  $M "$@" > /tmp/one.txt 2> /tmp/one.err || { echo "single run failed: $*"; cat /tmp/one.err; exit 1; }
  for n in 2 3 4 7; do
    timeout -k 5 120 $M "$@" -gpus $n > /tmp/n.txt 2> /tmp/n.err || { echo "run with $n slabs failed: $*"; tail -5 /tmp/n.err; exit 1; }
    if cmp -s /tmp/one.txt /tmp/n.txt; then echo "ok   $n slabs: $*   [$(sed -n 2p /tmp/n.err)]"; else echo "DIFF $n slabs: $*"; fail=1; fi
  done
}
cmp_run -nx 256 -ny 203 -w 0.8 -s 1001 -of 100 -omv
cmp_run -nx 256 -ny 203 -w 0.8 -s 1001 -of 100 -omv -push
cmp_run -nx 96 -ny 70 -w 1.2 -s 201 -of 100 -ovf -ld
cmp_run -nx 64 -ny 61 -w 1.0 -s 50 -of 7 -ov
cmp_run -nx 1024 -ny 2048 -w 0.5 -s 300 -of 50 -omv
exit $fail
