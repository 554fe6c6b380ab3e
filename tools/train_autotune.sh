#!/bin/bash
# Runs ON THE GPU BOX: train.py --conv-mode autotune with MIOpen's NORMAL find (every applicable solver compiled and timed
# at the first call of each convolution shape), into a fresh MIOpen user cache that is packed afterwards; then a second
# process on the warm cache with phase brackets.  Usage: gpurun -- 'bash tools/train_autotune.sh r03 c2|c4 [FIND_MODE]'
set -u
TAG=${1:-r03}; CFG=${2:-c2}; MODE=${3:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
M=/tmp/miopen_autotune_$CFG; mkdir -p $OUT $M/db $M/cache
export MIOPEN_USER_DB_PATH=$M/db MIOPEN_CUSTOM_CACHE_DIR=$M/cache MIOPEN_FIND_MODE=$MODE
cd $R
case $CFG in c2) flags="--batch 8";; c4) flags="--model multi --views 5 --batch 16";; esac
SECONDS=0
timeout 2400 python3 train.py $flags --conv-mode autotune --steps 6 --warmup 3 > $OUT/${TAG}_autotune_${CFG}_cold.json 2> $OUT/${TAG}_autotune_${CFG}.err
echo "{\"cold_process_wall_s\": $SECONDS, \"MIOPEN_FIND_MODE\": $MODE}" >> $OUT/${TAG}_autotune_${CFG}_cold.json
timeout 900 python3 train.py $flags --conv-mode autotune --steps 20 --warmup 5 --phase-times > $OUT/${TAG}_autotune_${CFG}.json 2>> $OUT/${TAG}_autotune_${CFG}.err
tail -n 1 $OUT/${TAG}_autotune_${CFG}.json | cut -c1-900
tail -n 1 $OUT/${TAG}_autotune_${CFG}_cold.json
du -sh $M/db $M/cache
[ $(du -sm $M | tail -1 | cut -f1) -lt 40 ] && tar czf $OUT/${TAG}_autotune_${CFG}_miopen_cache.tgz -C $M .
