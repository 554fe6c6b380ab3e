#!/bin/bash
# Builds an experiment variant of libsvbrdf_hip.so into tools/_build/libsvbrdf_<tag>.so with extra flags for both
# translation units (e.g. -DSVBRDF_X=1) and optional overrides SCHED_MAIN=... SCHED_ADJOINT=... in the environment.
#   bash tools/build_variant.sh <tag> [extra compiler flags...]
# SRC_DIR=<dir holding svbrdf_kernels.hip and svbrdf_hip.h> builds another revision of the source (e.g. files taken
# with `git show <rev>:...`) for a same-box A/B against it.
set -e
tag=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
src=${SRC_DIR:-$root/svbrdf_estimation_amd/csrc}
inc=${SRC_DIR:-$root/include}
out=$root/tools/_build
mkdir -p $out/obj_$tag
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -std=c++17 -fvisibility=hidden -I$inc $*"
SM=${SCHED_MAIN:--mllvm -misched-postra-direction=bottomup}
SA=${SCHED_ADJOINT:--mllvm -enable-post-misched=0 -mllvm -amdgpu-sched-strategy=iterative-minreg -mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -greedy-reverse-local-assignment}
SX=${SCHED_ADJOINT_EXTRA:-$SA}
/opt/rocm/bin/hipcc $F $SM -DSVBRDF_TU=0 -c -o $out/obj_$tag/main.o $src/svbrdf_kernels.hip &
/opt/rocm/bin/hipcc $F $SA -DSVBRDF_TU=1 -c -o $out/obj_$tag/adj.o $src/svbrdf_kernels.hip &
/opt/rocm/bin/hipcc $F $SX -DSVBRDF_TU=3 -c -o $out/obj_$tag/adjx.o $src/svbrdf_kernels.hip &
AUX=""
if [ -f $src/svbrdf_aux_f64.hip ]; then      # round 5 on: the auxiliary float64 unit is a file of its own
  /opt/rocm/bin/hipcc $F -c -o $out/obj_$tag/aux.o $src/svbrdf_aux_f64.hip &
  AUX=$out/obj_$tag/aux.o
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $out/libsvbrdf_$tag.so $out/obj_$tag/main.o $out/obj_$tag/adj.o $out/obj_$tag/adjx.o $AUX
echo built $out/libsvbrdf_$tag.so
