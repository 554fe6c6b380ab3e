#!/bin/bash
# Builds timing-only ablation variants of the kernels (WRONG results by construction) and, on the GPU
# box, times the fused loss with each:   bash tools/ablate.sh build   |   bash tools/ablate.sh run
set -e
cd "$(dirname "$0")/.."
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -fvisibility=hidden -Iinclude -shared -mllvm -enable-post-misched=0 -mllvm -amdgpu-sched-strategy=max-memory-clause"
if [ "$1" = build ]; then
  mkdir -p tools/_build
  for n in 1 2 3 4 5 6 7; do /opt/rocm/bin/hipcc $FLAGS -DSVBRDF_ABLATE=$n -o tools/_build/libsvbrdf_ablate$n.so svbrdf_estimation_amd/csrc/svbrdf_kernels.hip; done
else
  export SWEEP=short SVBRDF_NO_HOST_EXT=1
  echo "== full kernel"; python tools/k3_sweep.py 2>&1 | grep "^B=" | grep -E "S=9 |S=18" | grep "grad=1"
  for n in 1 2 3 4 5 6 7; do echo "== ablation $n"; SVBRDF_HIP_LIB=$PWD/tools/_build/libsvbrdf_ablate$n.so python tools/k3_sweep.py 2>&1 | grep "^B=" | grep -E "S=9 |S=18" | grep "grad=1"; done
fi
