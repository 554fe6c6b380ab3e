#!/bin/bash
# PMC comparison of kernel builds on one box:  bash tools/pmc_ab.sh libA.so libB.so   (outputs under gpurun_out/pmc_ab/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  export SVBRDF_HIP_LIB=$R/$lib
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/${n}_p1 --output-format csv -- python3 $R/tools/k3_one.py > $OUT/${n}_p1.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM -d $OUT/${n}_p2 --output-format csv -- python3 $R/tools/k3_one.py > $OUT/${n}_p2.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU2 SQ_CYCLES -d $OUT/${n}_p3 --output-format csv -- python3 $R/tools/k3_one.py > $OUT/${n}_p3.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_ab")
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    lib = f.split("/pmc_ab/")[1].split("_p")[0]
    for r in csv.DictReader(open(f)):
        if "rendering_loss" in r["Kernel_Name"]:
            res[lib][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for l in res.values() for c in l})
libs = sorted(res)
print("%-26s" % "counter" + "".join("%18s" % l for l in libs))
for c in names:
    print("%-26s" % c + "".join("%18.0f" % (sum(res[l][c][2:]) / max(1, len(res[l][c][2:]))) if res[l][c] else "%18s" % "-" for l in libs))
PY
