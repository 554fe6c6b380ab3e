#!/bin/bash
# Round-4 evidence run on the GPU box:  gpurun -- 'bash tools/round4_gpu.sh'
#   GPU test suite (with the allowance ledger), same-box A/B of the K3 builds, occupancy timelines, rocprofv3 stats and
#   PMC passes + un-profiled bench line (collect_profiles.sh), the driver's short bench form.
# Afterwards, here: python tools/summarize_profiles.py r04; copy the r04_* text files from gpurun_out/ to profiles/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "^\[datapath\]|^rank placement|over RCCL|two ranks on one device|DDP \(2 ranks\)|config 4 end to end|ledger:|passed|failed|Error" > gpurun_out/r04_gputest.txt
tail -4 gpurun_out/r04_gputest.txt
cp gpurun_out/tolerance_uses.txt gpurun_out/r04_tolerance_uses.txt 2>/dev/null
if [ -z "$SKIP_AB" ]; then
  bash tools/round4_ab.sh > gpurun_out/r04_k3_ab.txt 2>&1
  for g in 1 3; do SVBRDF_HIP_LIB=$PWD/tools/_build/libsvbrdf_r4tim.so SVBRDF_K3_SPLIT=$g python tools/k3_timeline.py 2>/dev/null > gpurun_out/r04_k3_timeline_split$g.txt; done
fi
bash tools/collect_profiles.sh r04
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_short.json 2> gpurun_out/r04_bench_short.err
