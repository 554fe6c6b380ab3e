#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's evidence -- GPU test suite + allowance ledger, the driver's short bench form
# twice, training-harness lines, rocprofv3 passes, the un-profiled bench line, the speed guard as a script.  SKIP_SUITE=1: the
# rocprofv3 / bench part only.
# Usage: gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r05'
# Outputs land in gpurun_out/<tag>_*; tools/summarize_profiles.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ -z "${SKIP_SUITE:-}" ]; then
  # the GPU test suite with its printed evidence lines, and the allowance ledger of that session
  python3 -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "^\[datapath\]|^\[perf-guard\]|^rank placement|over RCCL|two ranks on one device|DDP \(2 ranks\)|config 4 end to end|ledger:|passed|failed|Error" > $OUT/${TAG}_gputest.txt
  tail -3 $OUT/${TAG}_gputest.txt
  cp $OUT/tolerance_uses.txt $OUT/${TAG}_tolerance_uses.txt 2>/dev/null
  # the driver's short form, twice on this box (`value` = the median of nine 20-step regions: the two must agree within 2 %)
  python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_short.json 2> $OUT/${TAG}_bench_short.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_short2.json 2> $OUT/${TAG}_bench_short2.err
  # training harness lines: RCCL world of one, and two DDP ranks sharing the device (the self-checking no_sync probe)
  python3 train.py --gpus 1 --force-dist --steps 10 --warmup 3 --batch 8 > $OUT/${TAG}_train_rccl_world1.json 2>/dev/null
  python3 train.py --gpus 2 --backend gloo --share-device --steps 10 --warmup 3 --batch 8 > $OUT/${TAG}_train_2ranks_share_device.json 2>/dev/null
fi
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-secondary --timed-only"
# kernel-trace + stats (no counters in these passes): the default command (every step on one stream: the duration of a
# launch on its own) and the same with the steps alternating on two streams (launches overlap, so the per-launch duration
# exceeds a launch's share of the GPU: what bench.py reports as two_streams_overlapped)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats --output-format csv -- $BENCH --steps 400 --warmup 50 > $OUT/${TAG}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats2 --output-format csv -- $BENCH --steps 400 --warmup 50 --streams 2 > $OUT/${TAG}_stats2.log 2>&1
# ... and of the DEFAULT command as it is (2,000 steps, every untimed follow-up leg included): the verbatim average then mixes
# in the overlapped launches of the two-stream leg, so the summary also splits the per-dispatch trace by stream
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_statsdef --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $OUT/${TAG}_statsdef.log 2>&1
# the counter passes run one launch at a time (--streams 1) over the six rotating batches (302 MB: beyond the Infinity Cache)
BENCH="$BENCH --streams 1"
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_write.log 2>&1
# SQ issue/stall counters
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d $OUT/${TAG}_pmc_sq --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM -d $OUT/${TAG}_pmc_sq2 --output-format csv -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_pmc_sq2.log 2>&1
# the other kernels and K3's other variants, one process per case (tools/kernel_cases.py): kernel-trace stats, then
# FETCH_SIZE and WRITE_SIZE in their own passes -- the HBM-bound K1 / K2 / K4 and the loss variants the headline does not run
for c in ${CASES:-k1 k2 k4 photos copy synthesis k3_mixed k3_head k3_head_l1 k3_untied k3_config4 k3_config5}; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_case_${c}_stats --output-format csv -- python3 $R/tools/kernel_cases.py $c 60 > $OUT/${TAG}_case_${c}.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_case_${c}_fetch --output-format csv -- python3 $R/tools/kernel_cases.py $c 12 >> $OUT/${TAG}_case_${c}.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_case_${c}_write --output-format csv -- python3 $R/tools/kernel_cases.py $c 12 >> $OUT/${TAG}_case_${c}.log 2>&1
  # the raw per-dispatch traces are large: keep the stats and the counter tables only
  find $OUT/${TAG}_case_${c}_stats $OUT/${TAG}_case_${c}_fetch $OUT/${TAG}_case_${c}_write -name '*_kernel_trace.csv' -delete 2>/dev/null
done
# VALU instruction counts of the K3 shapes the headline does not run (their VALU-issue fraction: DESIGN.md section 4)
for c in ${SQ_CASES:-k3_untied k3_config4 k3_config5}; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES -d $OUT/${TAG}_case_${c}_sq --output-format csv -- python3 $R/tools/kernel_cases.py $c 12 >> $OUT/${TAG}_case_${c}.log 2>&1
  find $OUT/${TAG}_case_${c}_sq -name '*_kernel_trace.csv' -delete 2>/dev/null
done
cd $R
# un-profiled bench line (with the CPU baseline) -- never compare profiled and un-profiled timings
timeout 400 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# back-to-back launches through the C ABI, cycles at the clock under load (the GPU suite's speed guard, as a script)
PERF_GUARD_REPEATS=2 timeout 200 python3 tests/test_gpu_perf_guard.py > $OUT/${TAG}_perf_guard.txt 2>&1 && cp $OUT/perf_guard.json $OUT/${TAG}_perf_guard.json
timeout 60 python3 tools/clock_timeline.py > $OUT/${TAG}_clock_timeline.txt 2>&1
# the copy kernel's A/B grid (the measured-copy peak bench.py prices the HBM-bound kernels against) and the placement
# check of K3's load stagger (timing builds of tools/build_variant.sh, when they travelled with the snapshot)
timeout 300 python3 tools/copy_peak.py > $OUT/${TAG}_copy_peak.txt 2>&1
for v in tim tim0; do
  [ -f tools/_build/libsvbrdf_$v.so ] && SVBRDF_HIP_LIB=tools/_build/libsvbrdf_$v.so timeout 120 python3 tools/k3_placement.py > $OUT/${TAG}_k3_placement_$v.txt 2>&1
done
# first contact self-test, as far as one GPU goes: RCCL world 1, and eight self-spawned ranks sharing the device over gloo
timeout 200 python3 bench.py --gpus 1 --force-dist --selftest > $OUT/${TAG}_selftest_rccl_world1.json 2> $OUT/${TAG}_selftest_rccl_world1.err
timeout 400 python3 bench.py --gpus 8 --backend gloo --share-device --selftest > $OUT/${TAG}_selftest_8ranks_share_device.json 2> $OUT/${TAG}_selftest_8ranks_share_device.err
[ -x tools/_build/valu_rate ] && timeout 100 tools/_build/valu_rate > $OUT/${TAG}_valu_rate.txt 2>&1
tail -c 600 $OUT/${TAG}_bench.json
