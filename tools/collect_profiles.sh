#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench + rocprofv3 passes for the round's evidence.
# Usage: gpurun -- 'bash tools/collect_profiles.sh r01'
# Outputs land in gpurun_out/<tag>_*; tools/summarize_profiles.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-secondary"
# kernel-trace + stats (no counters in these passes): the default command (steps on two streams: launches overlap, so the
# per-launch duration exceeds a launch's share of the GPU) and the same with every step on one stream (the duration of a
# launch on its own: what bench.py reports as roofline.one_launch_alone / single_stream)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats --output-format csv -- $BENCH --steps 400 --warmup 50 > $OUT/${TAG}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats1 --output-format csv -- $BENCH --steps 400 --warmup 50 --streams 1 > $OUT/${TAG}_stats1.log 2>&1
# the counter passes run one launch at a time (--streams 1) over the six rotating batches (302 MB: beyond the Infinity Cache)
BENCH="$BENCH --streams 1"
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_write.log 2>&1
# SQ issue/stall counters
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d $OUT/${TAG}_pmc_sq --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM -d $OUT/${TAG}_pmc_sq2 --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_sq2.log 2>&1
cd $R
# un-profiled bench line (with the CPU baseline) -- never compare profiled and un-profiled timings
timeout 400 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 200 python3 tools/k3_sweep.py > $OUT/${TAG}_k3_sweep.txt 2>&1
timeout 200 python3 tools/k12_bench.py > $OUT/${TAG}_k12_bench.txt 2>&1
[ -x tools/_build/valu_rate ] && timeout 100 tools/_build/valu_rate > $OUT/${TAG}_valu_rate.txt 2>&1
tail -c 600 $OUT/${TAG}_bench.json
