#!/usr/bin/env python3
"""Markdown tables of a tools/round4_ab.sh output file (profiles/r04_k3_ab.txt): part 1 (cache-resident maps, second
round) and part 2 (maps from HBM, median of three rounds), one row per library build -- what profiles/HISTORY.md quotes."""
import collections
import re
import sys

LABELS = collections.OrderedDict([
    ("r3", "round-3 kernels"), ("r4", "**round 4, shipped**"),
    ("split=2", "scene split, 64 pixels × 2 waves"), ("split=3", "scene split × 3 (the verdict's variant)"), ("split=4", "scene split × 4"),
    ("r4t64", "64-thread workgroups (unsplit)"), ("r4t128", "128-thread workgroups"),
    ("r4peel", "last render of a wave without the unused successor geometry (peel)"),
    ("r4nopipe", "geometry of a render in its own pass (no unused geometry, no interleave)"),
    ("r4plain", "gradient stores without a cache policy (round 3)"), ("r4nt", "… `nt`"), ("r4ntwt", "… `sc0 sc1 nt`"),
    ("r4late", "coordinates loaded after the planes (round 3's order)"),
    ("r4nostag", "first round's plane loads all at once (round 3)"),
    ("r4dot3", "`n·wo`, `n·wi`, `wo·h` as three rounded products + two adds (round 3)"),
    ("r4nodef", "1/π and 4r³ of the gradient per render and channel (round 3)"),
    ("r4nolerp", "`f = (1−F)d/π + F·GD` as two products (round 3)"), ("r4alg0", "none of the three instruction-count reductions"),
    ("r4p1", "`s_setprio` by remaining renders, last resident round only"), ("r4p2", "… every wave")])


def main(path):
    part1, part2, part, rnd = collections.OrderedDict(), collections.OrderedDict(), 1, 0
    for line in open(path):
        if line.startswith("== round"):
            part, rnd = 1, int(line.split()[2])
        elif line.startswith("== hbm round"):
            part = 2
        m = re.match(r"^(\S+)\s+(\S+)\s+([\d.]+) us\s+([\d.]+) us", line)
        if not m:
            continue
        cfg, lib, a, b = m.group(1), m.group(2), float(m.group(3)), float(m.group(4))
        if part == 1 and rnd == 2:
            part1.setdefault(lib, {})[cfg] = (a, b)
        elif part == 2:
            part2.setdefault(lib, {}).setdefault(cfg, []).append((a, b))
    cfgs = ["tied", "untied", "mixed", "head+l1", "tied-B16", "tied-B4"]
    print("| build | tied | untied | mixed | head + L1 | B = 16 | B = 4 |\n|---|---|---|---|---|---|---|")
    for lib, label in LABELS.items():
        if lib in part1:
            print("| %s | " % label + " | ".join("%.2f / %.2f" % part1[lib][c] if c in part1[lib] else "" for c in cfgs) + " |")
    print("\n| build, maps from HBM (median of three rounds) | tied | untied | mixed | B = 16 |\n|---|---|---|---|---|")
    med = lambda v, i: sorted(x[i] for x in v)[len(v) // 2]
    for lib, label in LABELS.items():
        if lib in part2:
            print("| %s | " % label + " | ".join("%.2f / %.2f" % (med(part2[lib][c], 0), med(part2[lib][c], 1))
                                                 for c in ["tied", "untied", "mixed", "tied-B16"]) + " |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_k3_ab.txt")
