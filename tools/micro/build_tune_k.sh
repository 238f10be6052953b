#!/bin/bash
# build_tune_k.sh TAG [extra -D flags...]: one binary of tools/micro/tune_k.hip per setting
TAG=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I lbm_amd/csrc -I include \
   -DTK_TAG="\"$TAG\"" "$@" tools/micro/tune_k.hip -o tools/micro/tune_k_$TAG
