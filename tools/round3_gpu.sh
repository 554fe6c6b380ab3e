#!/bin/bash
# Round-3 evidence run ON THE GPU BOX: GPU test suite, same-box A/B of kernel builds, the profile collection, and the
# trace of reference-shaped render() calls.   gpurun -- 'bash tools/round3_gpu.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q -s > $OUT/r03_gpu_tests2.txt 2>&1; tail -4 $OUT/r03_gpu_tests2.txt
bash tools/ab_libs.sh tools/_build/libsvbrdf_base.so tools/_build/libsvbrdf_advice.so tools/_build/libsvbrdf_w5.so > $OUT/r03_ab_advice.txt 2>&1; cat $OUT/r03_ab_advice.txt
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/r03_render_calls --output-format csv -- python3 $R/tools/render_call_trace.py > $OUT/r03_render_calls.log 2>&1 ); tail -2 $OUT/r03_render_calls.log
bash tools/collect_profiles.sh r03
