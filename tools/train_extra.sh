#!/bin/bash
# Runs ON THE GPU BOX: the remaining BASELINE training shapes through train.py (--conv-mode hybrid, the default):
#   c3  config 3's per-GPU shape: Deschaintre tiled-PNG samples through the real reader + DataLoader workers, batch 8
#   c5  config 5's per-GPU shape: 512x512, 11 + 21 scenes, batch 8, mixed loss
#   c2h config 2 with the network head folded into the loss kernel (--fused-head)
# Usage: gpurun -- 'bash tools/train_extra.sh r03'
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python3 tools/make_png_dataset.py /tmp/pngds --samples 256 --tile 288 --photos 1 > $OUT/${TAG}_pngds.log 2>&1
for w in 2 8 16; do
  timeout 900 python3 train.py --data /tmp/pngds --image-count 1 --random-crop --batch 8 --steps 40 --warmup 10 --workers $w --phase-times > $OUT/${TAG}_train_c3_png_w$w.json 2> $OUT/${TAG}_train_c3_png_w$w.err
  tail -n 1 $OUT/${TAG}_train_c3_png_w$w.json | cut -c1-200
done
timeout 900 python3 train.py --data /tmp/pngds --image-count 1 --random-crop --batch 8 --steps 60 --warmup 10 --workers 16 > $OUT/${TAG}_train_c3_png.json 2>> $OUT/${TAG}_train_c3_png_w16.err
SECONDS=0
timeout 1500 python3 train.py --size 512 --random-scenes 11 --specular-scenes 21 --batch 8 --steps 6 --warmup 5 > $OUT/${TAG}_train_c5_cold.json 2> $OUT/${TAG}_train_c5.err
echo "{\"cold_process_wall_s\": $SECONDS}" >> $OUT/${TAG}_train_c5_cold.json
timeout 900 python3 train.py --size 512 --random-scenes 11 --specular-scenes 21 --batch 8 --steps 10 --warmup 5 --phase-times > $OUT/${TAG}_train_c5.json 2>> $OUT/${TAG}_train_c5.err
tail -n 1 $OUT/${TAG}_train_c5.json | cut -c1-700
timeout 900 python3 train.py --fused-head --batch 8 --steps 20 --warmup 5 --phase-times > $OUT/${TAG}_train_c2_fused_head.json 2> $OUT/${TAG}_train_c2_fused_head.err
tail -n 1 $OUT/${TAG}_train_c2_fused_head.json | cut -c1-700
timeout 900 python3 train.py --batch 8 --steps 20 --warmup 5 --phase-times > $OUT/${TAG}_train_c2_again.json 2> $OUT/${TAG}_train_c2_again.err
tail -n 1 $OUT/${TAG}_train_c2_again.json | cut -c1-700
tar czf $OUT/${TAG}_miopen_cache_full2.tgz -C svbrdf_estimation_amd/training/miopen_cache .
