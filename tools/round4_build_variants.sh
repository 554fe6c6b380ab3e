#!/bin/bash
# Builds the library variants tools/round4_ab.sh compares (here, on the CPU box: hipcc cross-compiles gfx950; the .so
# files travel to the GPU box with the snapshot).  r3 = the round-3 kernel source (git show <rev>:...).
set -e
cd "$(dirname "$0")/.."
R3=${R3_REV:-1dc6d72}
mkdir -p /tmp/r3src
git show $R3:svbrdf_estimation_amd/csrc/svbrdf_kernels.hip > /tmp/r3src/svbrdf_kernels.hip
git show $R3:include/svbrdf_hip.h > /tmp/r3src/svbrdf_hip.h
SRC_DIR=/tmp/r3src bash tools/build_variant.sh r3
bash tools/build_variant.sh r4                                         # the shipped source and flags
bash tools/build_variant.sh r4split -DSVBRDF_K3_SPLIT_VARIANTS=1       # + scene-split layouts (SVBRDF_K3_SPLIT=2|3|4)
bash tools/build_variant.sh r4t64 -DSVBRDF_K3_THREADS=64
bash tools/build_variant.sh r4t128 -DSVBRDF_K3_THREADS=128
bash tools/build_variant.sh r4peel -DSVBRDF_K3_PEEL_LAST=1
bash tools/build_variant.sh r4plain -DSVBRDF_K3_STORE_AUX=0            # gradient stores without a cache policy (round 3)
bash tools/build_variant.sh r4nt -DSVBRDF_K3_STORE_AUX=2
bash tools/build_variant.sh r4ntwt -DSVBRDF_K3_STORE_AUX=19
bash tools/build_variant.sh r4late -DSVBRDF_K3_EARLY_COORDS=0          # coordinates loaded after the planes (round 3)
bash tools/build_variant.sh r4dot3 -DSVBRDF_FMA_VN_LN=0                # n.wo, n.wi, wo.h as three rounded products + two adds (round 3)
bash tools/build_variant.sh r4nodef -DSVBRDF_K3_DEFER_SCALES=0         # 1/pi and 4r^3 of the gradient per render and channel (round 3)
bash tools/build_variant.sh r4nolerp -DSVBRDF_F_AS_LERP=0              # f = (1-F) d/pi + F GD as two products (round 3)
bash tools/build_variant.sh r4alg0 -DSVBRDF_FMA_VN_LN=0 -DSVBRDF_K3_DEFER_SCALES=0 -DSVBRDF_F_AS_LERP=0   # none of the three
bash tools/build_variant.sh r4nopipe -DSVBRDF_K3_PIPELINE=0            # geometry of a render in its own pass
bash tools/build_variant.sh r4p1 -DSVBRDF_K3_TAIL_PRIO=1
bash tools/build_variant.sh r4p2 -DSVBRDF_K3_TAIL_PRIO=2
bash tools/build_variant.sh r4nostag -DSVBRDF_K3_STAGGER=0              # first round's plane loads all at once (round 3)
bash tools/build_variant.sh r4tim -DSVBRDF_TIMING=1 -DSVBRDF_K3_SPLIT_VARIANTS=1
hipcc -O2 --offload-arch=gfx950 -Iinclude tools/k3_split_bench.cpp -o tools/_build/k3_split_bench -ldl
