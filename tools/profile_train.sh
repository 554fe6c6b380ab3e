#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): one training step of train.py under the clock, with phase brackets, and under
# rocprofv3 --kernel-trace --stats, at BASELINE config 2 (single view, batch 8) and configs[3] (multi-view N=5, batch 16).
# Usage: gpurun -- 'bash tools/profile_train.sh r03 [extra train.py flags]'
# The MIOpen caches (kernel binaries + find results compiled at the first step: the image ships no gfx950 database) are
# kept in one directory so that the later processes of this call, and their size, are visible afterwards.
set -u
TAG=${1:-r03}; shift || true
EXTRA="$*"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
M=/tmp/miopen_r03
mkdir -p $OUT $M/db $M/cache
export MIOPEN_USER_DB_PATH=$M/db MIOPEN_CUSTOM_CACHE_DIR=$M/cache
cd /tmp && export TMPDIR=/tmp
T="python3 $R/train.py $EXTRA"
for cfg in "c2:--batch 8" "c4:--model multi --views 5 --batch 16"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  # 1. cold process (compiles MIOpen's kernels at the first steps), wall time of the whole process recorded
  SECONDS=0; timeout 1500 $T $flags --steps 6 --warmup 5 > $OUT/${TAG}_train_${name}_cold.json 2> $OUT/${TAG}_train_${name}_cold.err
  echo "cold process wall seconds: $SECONDS" > $OUT/${TAG}_train_${name}_cold.time
  # 2. warm process: throughput, then phases
  timeout 600 $T $flags --steps 10 --warmup 5 > $OUT/${TAG}_train_${name}.json 2> $OUT/${TAG}_train_${name}.err
  timeout 600 $T $flags --steps 6 --warmup 5 --phase-times > $OUT/${TAG}_train_${name}_phases.json 2>> $OUT/${TAG}_train_${name}.err
  # 3. kernel trace (dataloader in-process: no forked workers under the profiler)
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_train_${name}_prof --output-format csv -- $T $flags --steps 4 --warmup 5 --workers 0 > $OUT/${TAG}_train_${name}_prof.log 2>&1
done
( du -sh $M/db $M/cache; find $M -type f | xargs ls -la ) > $OUT/${TAG}_miopen_cache_size.txt 2>&1
# the caches travel back only when they are small (gpurun merges at most 64 MiB)
[ $(du -sm $M | tail -1 | cut -f1) -lt 40 ] && tar czf $OUT/${TAG}_miopen_cache.tgz -C $M . 
cat $OUT/${TAG}_train_c2.json $OUT/${TAG}_train_c4.json $OUT/${TAG}_miopen_cache_size.txt
