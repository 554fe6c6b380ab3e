#!/bin/bash
# Runs ON THE GPU BOX: bounded MIOpen find over configs[3]'s convolution shapes (80 images), forward first, then forward +
# backward, each under its own time limit; the in-tree cache is packed after every stage.  gpurun --timeout 2400 -- 'bash tools/search_c4.sh r03 900 900'
set -u
TAG=${1:-r03}; T1=${2:-900}; T2=${3:-900}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
D=$R/svbrdf_estimation_amd/training/miopen_cache
mkdir -p $OUT $D/db $D/cache
pack() { tar czf $OUT/${TAG}_miopen_cache_c4search.tgz -C $D . && ls -la $OUT/${TAG}_miopen_cache_c4search.tgz; }
trap pack EXIT
cd $R
export MIOPEN_FIND_MODE=1
SECONDS=0; timeout -s INT $T1 python3 tools/conv_search.py multi 16 5 fwd 2>&1 | grep -v amdgpu.ids | tail -3; echo "forward search: ${SECONDS}s"; pack
SECONDS=0; timeout -s INT $T2 python3 tools/conv_search.py multi 16 5 fwdbwd 2>&1 | grep -v amdgpu.ids | tail -3; echo "forward+backward search: ${SECONDS}s"; pack
unset MIOPEN_FIND_MODE
timeout 600 python3 train.py --model multi --views 5 --batch 16 --steps 10 --warmup 5 --phase-times 2>/dev/null | tail -1 | cut -c1-900 | tee $OUT/${TAG}_c4_after_search.json
