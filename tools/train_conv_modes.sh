#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): train.py at config 2 and configs[3] under the ways stock PyTorch-ROCm can pick the
# U-Net's MIOpen convolution algorithms (--conv-mode), cold (kernel compilation included) and warm.
# Usage: gpurun -- 'bash tools/train_conv_modes.sh r03 "fast" "autotune:MIOPEN_FIND_MODE=2" "fast_cl" ...'
#   spec = mode[_cl][:ENV=value ...]   (_cl adds --channels-last)
set -u
TAG=${1:-r03}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  mode=${spec%%:*}; envs=""; [ "$spec" != "$mode" ] && envs=${spec#*:}
  name=$(echo "$spec" | tr -c 'A-Za-z0-9_\n' '_')
  M=/tmp/miopen_$name; mkdir -p $M/db $M/cache
  extra=""
  case $mode in *_cl) extra="--channels-last";; esac
  m=${mode%_cl}
  for cfg in "c2:--batch 8" "c4:--model multi --views 5 --batch 16"; do
    c=${cfg%%:*}; flags=${cfg#*:}
    SECONDS=0
    env MIOPEN_USER_DB_PATH=$M/db MIOPEN_CUSTOM_CACHE_DIR=$M/cache $envs timeout 900 python3 $R/train.py $flags --conv-mode $m $extra --steps 6 --warmup 5 --workers 0 \
        > $OUT/${TAG}_mode_${name}_${c}_cold.json 2> $OUT/${TAG}_mode_${name}_${c}.err
    echo "{\"cold_process_wall_s\": $SECONDS}" >> $OUT/${TAG}_mode_${name}_${c}_cold.json
    env MIOPEN_USER_DB_PATH=$M/db MIOPEN_CUSTOM_CACHE_DIR=$M/cache $envs timeout 600 python3 $R/train.py $flags --conv-mode $m $extra --steps 10 --warmup 5 --workers 0 --phase-times \
        > $OUT/${TAG}_mode_${name}_${c}.json 2>> $OUT/${TAG}_mode_${name}_${c}.err
    echo "$spec $c: $(tail -n 1 $OUT/${TAG}_mode_${name}_${c}.json | cut -c1-330) cold ${SECONDS}s"
  done
  du -sh $M/db $M/cache > $OUT/${TAG}_mode_${name}_cache_size.txt 2>&1
done
