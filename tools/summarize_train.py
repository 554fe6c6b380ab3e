#!/usr/bin/env python3
"""Turns the raw output of tools/profile_train.sh / tools/train_conv_modes.sh merged into gpurun_out/ into the committed
evidence under profiles/:
   profiles/<tag>_train_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of train.py (top 25 kernels)
   profiles/<tag>_train_summary.json              per configuration: throughput, phase means, cold-process wall time
   profiles/<tag>_train_conv_modes.json           the --conv-mode comparison (phase means per mode)
Usage: python3 tools/summarize_train.py r03 [r03h ...]"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def last_json(path):
    if not os.path.exists(path):
        return None
    rows = [json.loads(l) for l in open(path) if l.startswith("{")]
    return rows[-1] if rows else None


for tag in sys.argv[1:] or ["r03"]:
    summary = {}
    for cfg in ("c2", "c4"):
        entry = {}
        run, ph = last_json(os.path.join(G, "%s_train_%s.json" % (tag, cfg))), last_json(os.path.join(G, "%s_train_%s_phases.json" % (tag, cfg)))
        if run:
            entry.update(patches_per_s=run["value"], ms_per_step=run["ms_per_step"], config=run["config"])
        if ph:
            entry["phase_ms_mean"] = ph.get("phase_ms_mean")
        t = os.path.join(G, "%s_train_%s_cold.time" % (tag, cfg))
        if os.path.exists(t):
            entry["cold_process"] = open(t).read().strip() + " (9 steps incl. MIOpen kernel compilation)"
        st = sorted(glob.glob(os.path.join(G, "%s_train_%s_prof/*/*_kernel_stats.csv" % (tag, cfg))), key=os.path.getmtime)
        if st:
            rows = list(csv.DictReader(open(st[-1])))
            total = sum(float(r["TotalDurationNs"]) for r in rows)
            entry["kernel_time_total_ms_in_profiled_run"] = total / 1e6
            with open(os.path.join(P, "%s_train_%s_kernel_stats.csv" % (tag, cfg)), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
                for r in rows[:25]:
                    w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
            entry["loss_kernel_share_pct"] = sum(float(r["Percentage"]) for r in rows if "k_rendering_loss" in r["Name"])
            entry["top5"] = [[r["Name"][:80], float(r["Percentage"])] for r in rows[:5]]
        if entry:
            summary[cfg] = entry
    if summary:
        json.dump(summary, open(os.path.join(P, "%s_train_summary.json" % tag), "w"), indent=1)
    modes = {}
    for f in sorted(glob.glob(os.path.join(G, "%s_mode_*_c?.json" % tag))):
        name = os.path.basename(f)[len(tag) + 6:-5]
        d = last_json(f)
        cold = last_json(f.replace(".json", "_cold.json"))
        modes[name] = {"phase_ms_mean": (d or {}).get("phase_ms_mean"), "config": (d or {}).get("config"),
                       "finished": d is not None, "cold_process_wall_s": (cold or {}).get("cold_process_wall_s")}
    if modes:
        json.dump({"note": "train.py --conv-mode comparison (tools/train_conv_modes.sh), --workers 0: data_wait is the in-process "
                           "CPU synthetic dataset of that run and is NOT part of the comparison; compare forward / backward",
                   "modes": modes}, open(os.path.join(P, "%s_train_conv_modes.json" % tag), "w"), indent=1)
    print("wrote", [f for f in sorted(os.listdir(P)) if f.startswith(tag + "_train")])
