#!/bin/bash
# Runs ON THE GPU BOX: extends the in-tree MIOpen user cache (svbrdf_estimation_amd/training/miopen_cache/: compiled
# kernels + find results) by running what runs MIOpen convolutions here -- the GPU test suite, train.py at the BASELINE
# shapes in the default hybrid mode, and `--conv-mode autotune` with MIOpen's NORMAL find at config 2 (7 min of search, once;
# configs[3]'s 80-image shapes did not finish their search in 44 min and are left to the hybrid mode) -- and packs the
# result into gpurun_out/<tag>_miopen_cache_built.tgz for tools/install_miopen_cache.sh, whatever happens on the way.
#   gpurun --timeout 2400 -- 'bash tools/build_miopen_cache.sh r03'
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
D=$R/svbrdf_estimation_amd/training/miopen_cache
mkdir -p $OUT $D/db $D/cache
pack() { tar czf $OUT/${TAG}_miopen_cache_built.tgz -C $D . && ls -la $OUT/${TAG}_miopen_cache_built.tgz; }
trap pack EXIT
cd $R
python3 -m pytest tests -m gpu -q > $OUT/${TAG}_cache_tests.txt 2>&1; tail -1 $OUT/${TAG}_cache_tests.txt
for f in "--batch 8" "--model multi --views 5 --batch 16" "--size 512 --random-scenes 11 --specular-scenes 21 --batch 8" "--fused-head --batch 8"; do
  timeout 900 python3 train.py $f --steps 3 --warmup 1 > /dev/null 2>&1
done
pack
SECONDS=0
MIOPEN_FIND_MODE=1 timeout 1200 python3 train.py --batch 8 --conv-mode autotune --steps 6 --warmup 3 > $OUT/${TAG}_tuned_c2_cold.json 2> $OUT/${TAG}_tuned_c2.err
echo "{\"search_process_wall_s\": $SECONDS}" >> $OUT/${TAG}_tuned_c2_cold.json
pack
timeout 600 python3 train.py --batch 8 --conv-mode autotune --steps 20 --warmup 5 --phase-times > $OUT/${TAG}_tuned_c2_phases.json 2>> $OUT/${TAG}_tuned_c2.err
timeout 600 python3 train.py --batch 8 --conv-mode autotune --steps 30 --warmup 5 > $OUT/${TAG}_tuned_c2.json 2>> $OUT/${TAG}_tuned_c2.err
tail -n 1 $OUT/${TAG}_tuned_c2.json | cut -c1-200
# the default mode on the finished cache: do the stored find results change what `hybrid` runs?
timeout 600 python3 train.py --batch 8 --steps 30 --warmup 5 --phase-times > $OUT/${TAG}_hybrid_on_tuned_cache_c2.json 2>/dev/null; tail -n 1 $OUT/${TAG}_hybrid_on_tuned_cache_c2.json | cut -c1-700
timeout 600 python3 train.py --batch 8 --conv-mode fast --steps 30 --warmup 5 --phase-times > $OUT/${TAG}_fast_on_tuned_cache_c2.json 2>/dev/null; tail -n 1 $OUT/${TAG}_fast_on_tuned_cache_c2.json | cut -c1-700
python3 -m pytest tests -m gpu -q -k "multirank or training_harness or unet" 2>&1 | tail -2
du -sh $D/db $D/cache
