#!/bin/bash
# Same-box A/B of libsvbrdf_hip.so builds through the C ABI (tools/k3_split_bench: no Python, no autograd):
# back-to-back launches on one stream (mode 0) and full launches alternating on two free-running streams (mode 4),
# for the rendering loss (tied / untied roughness) and the mixed loss.  Usage on the GPU box: bash tools/ab_libs.sh libA.so libB.so ...
cd "$(dirname "$0")/.."
for round in 1 2; do
  for lib in "$@"; do
    for cfg in ${AB_CFGS:-"tied:" "untied:K3_UNTIED=1" "mixed:K3_L1=0.1" "head+l1:K3_HEAD=1,K3_L1=0.1"}; do
      tag=${cfg%%:*}; envs=${cfg#*:}
      printf "%-34s %-7s " "$(basename $lib)" "$tag"
      env ${envs//,/ } K3_LIB=$lib K3_MODES=04 K3_ROUNDS=1 K3_STEPS=600 tools/_build/k3_split_bench | awk '{printf "%s us  ", $(NF-5)} END {print ""}'
    done
  done
done
