#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of library builds (tools/_build/libsvbrdf_<tag>.so, tools/build_variant.sh) with the
# GPU suite's speed-guard harness (tests/test_gpu_perf_guard.py as a script: back-to-back launches through the C ABI,
# cycles at the clock read during them) -- any shape the harness knows, unlike k3_split_bench (config 2 only).  The builds
# are visited in turn within each round, so drift of the box hits them alike.
#   CASES=k3_config2,k3_config2_untied,k3_config5_shape ROUNDS=3 bash tools/ab_perf_guard.sh r5base r5onercp r5rowrl
cd "$(dirname "$0")/.."
CASES=${CASES:-k3_config2,k3_config2_untied,k3_config2_mixed,k3_config5_shape}
for round in $(seq 1 ${ROUNDS:-3}); do
  for t in "$@"; do
    SVBRDF_HIP_LIB=$PWD/tools/_build/libsvbrdf_$t.so SVBRDF_NO_HOST_EXT=1 PERF_GUARD_CASES=$CASES PERF_GUARD_REPEATS=1 \
      python3 tests/test_gpu_perf_guard.py 2>/dev/null | grep "^\[perf-guard\]" | sed "s/^\[perf-guard\]/round $round $t/"
  done
done
