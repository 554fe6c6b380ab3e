#!/bin/bash
# Same-box A/B of kernel builds (devices differ by several % in VALU-bound loops: never compare across
# gpurun calls).  Usage on the GPU box: bash tools/ab.sh libA.so libB.so ...   (3 interleaved rounds)
export SWEEP=short SVBRDF_NO_HOST_EXT=1
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in "$@"; do
    printf "%-28s " "$(basename $lib)"
    SVBRDF_HIP_LIB=$PWD/$lib python tools/k3_sweep.py 2>&1 | grep "^B=" | grep grad=1 | grep -E "S=9 |S=18" | awk '{printf "%s %s us   ", $3, $6}'; echo
  done
done
