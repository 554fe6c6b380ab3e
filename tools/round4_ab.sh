#!/bin/bash
# Round 4, VERDICT task 1: same-box A/B of K3 builds through the C ABI without Python (tools/k3_split_bench: first
# number = launches back to back on one stream, second = launches alternating on two streams; us per launch at
# config 2 -- B=8, 256x256, 9 scenes -- unless the tag says otherwise).  Libraries: tools/round4_build_variants.sh.
#   r3       the round-3 kernels                      r4       the shipped kernels
#   split=G  64 pixels x G waves per workgroup, the renders of a pixel shared among the waves (LDS combine)
#   wg64/128 64- / 128-thread workgroups of the shipped layout
#   peel     last render of a wave without the unused successor geometry
#   plain/nt/ntwt  cache policy of the gradient stores (shipped: sc0 sc1)
#   late     pixel coordinates loaded after the plane loads (round 3's order)
#   dot3 / nodef / nolerp / alg0   the shipped build WITHOUT one / all of round 4's instruction-count reductions:
#            n.wo, n.wi, wo.h as FMAs; 1/pi and 4r^3 of the gradient once per pixel; f as d/pi + F (GD - d/pi)
#   nopipe   geometry of a render in its own pass (no unused geometry after the last render, no interleave)
#   prio1/2  s_setprio by remaining renders: last resident round only / every wave
#   nostag   first round's plane loads all at once (shipped: four layers, 48 x 64 cycles apart; matters with maps from HBM)
# Part 2 repeats the comparison for the changes that depend on it with the maps coming from HBM (K3_ROTATE=6: six sets of
# maps, 453 MB, beyond the 256 MiB Infinity Cache -- what bench.py measures); part 1 keeps ONE set cache-resident.
# Usage on the GPU box: bash tools/round4_ab.sh > gpurun_out/r04_k3_ab.txt
cd "$(dirname "$0")/.."
B=tools/_build
one() {   # label lib env...
  label=$1; lib=$2; shift 2
  printf "%-10s %-8s " "$label" "$(basename $lib .so | sed s/libsvbrdf_//)"
  env "$@" K3_LIB=$lib K3_MODES=${MODES:-04} K3_ROUNDS=1 K3_STEPS=${K3_STEPS:-600} $B/k3_split_bench | awk '{printf "%s us  ", $(NF-5); l=$NF} END {print " loss " l}'
}
CFGS=("tied:" "untied:K3_UNTIED=1" "mixed:K3_L1=0.1" "head+l1:K3_HEAD=1,K3_L1=0.1" "tied-B16:K3_B=16" "tied-B4:K3_B=4")
for round in 1 2; do
  echo "== round $round"
  for cfg in "${CFGS[@]}"; do
    tag=${cfg%%:*}; envs=${cfg#*:}
    one "$tag" $B/libsvbrdf_r3.so ${envs//,/ }
    one "$tag" $B/libsvbrdf_r4.so ${envs//,/ }
    for g in 2 3 4; do one "$tag" $B/libsvbrdf_r4split.so SVBRDF_K3_SPLIT=$g ${envs//,/ } | sed "s/r4split /split=$g /"; done
    for v in dot3 nodef nolerp alg0 nopipe t64 t128 peel plain nt ntwt late nostag p1 p2; do one "$tag" $B/libsvbrdf_r4$v.so ${envs//,/ }; done
  done
  MODES=6 one "floor" $B/libsvbrdf_r4.so | sed 's/$/   (a kernel that exits at once, launched back to back)/'
done
echo "== part 2: maps from HBM (K3_ROTATE=6), median-friendly: three rounds"
for round in 1 2 3; do
  echo "== hbm round $round"
  for cfg in "tied:" "untied:K3_UNTIED=1" "mixed:K3_L1=0.1" "tied-B16:K3_B=16"; do
    tag=${cfg%%:*}; envs=${cfg#*:}
    for v in r3 r4 r4nostag r4plain r4late r4alg0; do one "$tag" $B/libsvbrdf_$v.so K3_ROTATE=6 ${envs//,/ }; done
  done
done
