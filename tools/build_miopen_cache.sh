#!/bin/bash
# Runs ON THE GPU BOX: extends the tracked MIOpen user cache (svbrdf_estimation_amd/training/miopen_cache/: compiled
# kernels + find results) by running what runs MIOpen convolutions here -- the GPU test suite and train.py at the
# BASELINE shapes in the default mode -- with MIOpen pointed at a SCRATCH COPY of the tracked files (the tracked files are
# never written: training.use_in_tree_miopen_cache), and packs the copy into gpurun_out/<tag>_miopen_cache_built.tgz for
# tools/install_miopen_cache.sh, whatever happens on the way.  No find/search budget goes here any more (the exhaustive
# find of round 3 is in the cache already): this only adds the kernels of shapes met since.
#   gpurun --timeout 1500 -- 'bash tools/build_miopen_cache.sh r05'
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
D=$R/svbrdf_estimation_amd/training/miopen_cache
W=/tmp/miopen_cache_work_$$
mkdir -p $OUT $W
cp -r $D/cache $D/db $W/
export MIOPEN_CUSTOM_CACHE_DIR=$W/cache MIOPEN_USER_DB_PATH=$W/db
pack() { rm -f $W/db/*.time; tar czf $OUT/${TAG}_miopen_cache_built.tgz -C $W . && ls -la $OUT/${TAG}_miopen_cache_built.tgz; du -sb $W/cache $W/db; }
trap pack EXIT
cd $R
python3 -m pytest tests -m gpu -q > $OUT/${TAG}_cache_tests.txt 2>&1; tail -1 $OUT/${TAG}_cache_tests.txt
for f in "--batch 8" "--model multi --views 5 --batch 16" "--size 512 --random-scenes 11 --specular-scenes 21 --batch 8" "--fused-head --batch 8"; do
  timeout 900 python3 train.py $f --steps 3 --warmup 1 > /dev/null 2>&1
done
# second pass of the suite: must not grow the copy any further (what a fresh box with the installed cache will see)
before=$(du -sb $W | cut -f1)
python3 -m pytest tests -m gpu -q -k "multirank or training_harness or unet or datapath" 2>&1 | tail -1
echo "cache bytes before / after the second pass: $before / $(du -sb $W | cut -f1)"
