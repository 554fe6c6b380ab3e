#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of the libraries tools/_build/libsvbrdf_<tag>.so named on the command line
# (tools/build_variant.sh builds them) through tools/k3_split_bench: us per launch, one launch at a time / launches
# alternating on two streams; part 1 with one cache-resident set of maps, part 2 with the maps from HBM (K3_ROTATE=6).
#   bash tools/ab_quick.sh r4 cand1 cand2 ... > gpurun_out/x.txt        (ROUNDS=3 by default)
cd "$(dirname "$0")/.."
B=tools/_build
one() {   # label lib env...
  label=$1; lib=$2; shift 2
  printf "%-10s %-8s " "$label" "$(basename $lib .so | sed s/libsvbrdf_//)"
  env "$@" K3_LIB=$lib K3_MODES=${MODES:-04} K3_ROUNDS=1 K3_STEPS=${K3_STEPS:-600} $B/k3_split_bench | awk '{printf "%s us  ", $(NF-5); l=$NF} END {print " loss " l}'
}
CFGS=(${CFGS:-"tied:" "untied:K3_UNTIED=1" "mixed:K3_L1=0.1" "head+l1:K3_HEAD=1,K3_L1=0.1"})
for part in cache hbm; do
  for round in $(seq 1 ${ROUNDS:-3}); do
    echo "== $part round $round"
    for cfg in "${CFGS[@]}"; do
      tag=${cfg%%:*}; envs=${cfg#*:}
      for t in "$@"; do
        if [ $part = cache ]; then one "$tag" $B/libsvbrdf_$t.so ${envs//,/ }; else one "$tag" $B/libsvbrdf_$t.so K3_ROTATE=6 ${envs//,/ }; fi
      done
    done
  done
done
