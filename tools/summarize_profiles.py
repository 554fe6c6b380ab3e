#!/usr/bin/env python3
"""Turns the raw rocprofv3 output merged into gpurun_out/ (tools/collect_profiles.sh) into the
small, committed evidence files under profiles/ :
   profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (verbatim)
   profiles/<tag>_pmc_summary.json     per-kernel means of the PMC passes + derived HBM traffic
   profiles/k3_hbm_traffic.json        what bench.py reports as roofline.traffic
   profiles/<tag>_*.txt / .json        sweep, K1/K2, VALU micro-benchmark, bench line
HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 64 B per 128-B request for coalesced streaming reads, so read bytes =
2 * FETCH_SIZE * 1024 -- confirmed on this kernel: 2*FETCH = 50.5 MB against 50.3 MB of
algorithmic reads; WRITE_SIZE is exact.
"""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)
    return f[-1] if f else None       # newest run


st = one("%s_stats/*/*_kernel_stats.csv" % tag)
if st:
    shutil.copy(st, os.path.join(P, "%s_kernel_stats.csv" % tag))                 # default command: two streams
st1 = one("%s_stats1/*/*_kernel_stats.csv" % tag)
if st1:
    shutil.copy(st1, os.path.join(P, "%s_kernel_stats_streams1.csv" % tag))      # --streams 1: one launch at a time
summary = {}
for p in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    f = one("%s_%s/*/*_counter_collection.csv" % (tag, p))
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        m = re.search(r"(k_\w+(?:<[^>]*>)?|__amd_\w+|vectorized_elementwise_kernel<[^>]*FillFunctor<\w+>)", k)
        short = m.group(1) if m else k[:60]
        summary.setdefault(short, {})[c] = {"mean": sum(v) / len(v), "n": len(v), "min": min(v), "max": max(v)}
k3 = next((k for k in summary if "k_rendering_loss" in k), None)
bench = None
bj = os.path.join(G, "%s_bench.json" % tag)
if os.path.exists(bj):
    lines = [l for l in open(bj) if l.startswith("{")]
    if lines:
        bench = json.loads(lines[-1])
        json.dump(bench, open(os.path.join(P, "%s_bench.json" % tag), "w"), indent=1)
if k3 and "FETCH_SIZE" in summary[k3] and "WRITE_SIZE" in summary[k3]:
    fetch_kib, write_kib = summary[k3]["FETCH_SIZE"]["mean"], summary[k3]["WRITE_SIZE"]["mean"]
    cfg = (bench or {}).get("config", {})
    traffic = {"kernel": k3, "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
               "read_bytes_corrected": 2.0 * fetch_kib * 1024.0, "write_bytes": write_kib * 1024.0,
               "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
               "correction": "gfx950: FETCH_SIZE x2 for coalesced streaming reads (MI355X_MICROARCH.md, HBM)",
               "B": cfg.get("global_batch"), "H": cfg.get("H"), "S": cfg.get("scenes"), "round": tag,
               "distinct_batches": cfg.get("distinct_batches"), "working_set_MiB": cfg.get("working_set_MiB"),
               "git_head": os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip() or "?"}
    for key, counter in (("valu_wave_instr_per_launch", "SQ_INSTS_VALU"), ("trans_wave_instr_per_launch", "SQ_INSTS_VALU_TRANS_F32")):
        if counter in summary[k3]:
            traffic[key] = summary[k3][counter]["mean"]
    json.dump(traffic, open(os.path.join(P, "k3_hbm_traffic.json"), "w"), indent=1)
    summary["_k3_traffic"] = traffic
json.dump(summary, open(os.path.join(P, "%s_pmc_summary.json" % tag), "w"), indent=1, sort_keys=True)
for name in ("k3_sweep.txt", "k12_bench.txt", "valu_rate.txt"):
    src = os.path.join(G, "%s_%s" % (tag, name))
    if os.path.exists(src):
        keep = [l for l in open(src) if "amdgpu.ids" not in l]
        open(os.path.join(P, "%s_%s" % (tag, name)), "w").writelines(keep)
print("wrote", sorted(os.listdir(P)))
