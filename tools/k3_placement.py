#!/usr/bin/env python3
"""Run-time check of what K3's load stagger assumes (svbrdf_kernels.hip, SVBRDF_K3_STAGGER): that the dispatcher places a
launch's workgroups BREADTH-FIRST -- workgroups 0..255 one per CU, 256..511 the CUs' second slots, and so on -- so that
"linear workgroup index >> 8" is "which of a CU's four workgroup slots", and the first four layers (1024 workgroups = 256
CUs x 4) are all resident from the start.  If placement were depth-first the stagger would be a pure delay.

Needs a -DSVBRDF_TIMING=1 build (per-wave stamps instead of gradients):
    bash tools/build_variant.sh tim -DSVBRDF_TIMING=1 [-DSVBRDF_K3_STAGGER=0]
    SVBRDF_HIP_LIB=tools/_build/libsvbrdf_tim.so python tools/k3_placement.py
Per layer: when its first / median / last workgroup entered the kernel (100 MHz stamps relative to the launch's first entry;
taken BEFORE the stagger's sleep, so they read the dispatcher, not the sleep), on how many distinct CUs it sits and how many
workgroups of the layer share a CU at most.  Exit code 1 when the first round is not breadth-first (the stagger then relies
on luck and should go)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SVBRDF_NO_HOST_EXT", "1")
import numpy as np      # noqa: E402
import torch            # noqa: E402

from bench import synthetic_maps                                  # noqa: E402
from svbrdf_estimation_amd import _native, environment            # noqa: E402

TICK_US = 0.01          # s_memrealtime: 100 MHz
M = 1 << 24
STAGGER_UNIT_US = 64 * 64 / 2.2e3       # s_sleep 64 = 4096 cycles; ~1.9 us at the ~2.2 GHz this kernel runs at


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1)
    B, H, S = 8, 256, 9
    sets = [(synthetic_maps(gen, B, H).to(dev), synthetic_maps(gen, B, H).to(dev)) for _ in range(6)]      # 300 MB: from HBM
    torch.manual_seed(0)
    table = environment.BatchSceneSampler(B, 3, 6).sample()
    for k in range(12):
        _native.rendering_loss(*sets[k % 6], table, 0.1, want_grad=True)
    torch.cuda.synchronize()
    verdicts = []
    for trial in range(5):
        torch.cuda.synchronize()
        _, g = _native.rendering_loss(*sets[trial % 6], table, 0.1, want_grad=True)
        torch.cuda.synchronize()
        g = g.cpu().numpy()
        # one wave = 64 consecutive pixels; one workgroup = 4 waves = 256 pixels; linear workgroup index = b * 256 + pix / 256
        entry = g[:, 3].reshape(B, -1)[:, ::64].astype(np.int64)          # [B, waves per item]
        where = g[:, 5].reshape(B, -1)[:, ::64].astype(np.int64)
        if (g[:, 5] != np.round(g[:, 5])).any() or where.max() >= (1 << 12):
            raise SystemExit("plane 5 does not hold placement stamps: is SVBRDF_HIP_LIB a -DSVBRDF_TIMING=1 build?")
        wg_entry = entry.reshape(B, -1, 4).min(axis=2).reshape(-1)          # first wave of each workgroup
        wg_where = where.reshape(B, -1, 4)[:, :, 0].reshape(-1)
        t0 = wg_entry.min()
        rel = ((wg_entry - t0) % M) * TICK_US
        n_layers = len(rel) // 256
        print("trial %d: %d workgroups, %d distinct CUs in all (xcc:se:sh:cu)" % (trial, len(rel), len(set(wg_where.tolist()))))
        print("  layer   first entry   median   last entry (us)   distinct CUs   max workgroups of the layer on one CU")
        rows = []
        for L in range(n_layers):
            r, w = rel[256 * L:256 * (L + 1)], wg_where[256 * L:256 * (L + 1)]
            counts = np.unique(w, return_counts=True)[1]
            rows.append((r.min(), np.median(r), r.max(), len(counts), counts.max()))
            print("  %5d   %10.2f   %6.2f   %10.2f        %8d   %8d" % ((L,) + rows[-1]))
        first_round = rows[:4]
        spread = max(r[0] for r in first_round) - min(r[0] for r in first_round)
        breadth = all(r[3] >= 250 and r[4] <= 2 for r in first_round)
        later = min(r[0] for r in rows[4:]) if len(rows) > 4 else float("nan")
        ok = breadth and spread <= STAGGER_UNIT_US
        verdicts.append(ok)
        print("  first round (layers 0-3): first entries within %.2f us of each other (one stagger unit = %.2f us); every layer on "
              ">= 250 distinct CUs with <= 2 of its workgroups per CU: %s; second round (layers 4+) starts at %.2f us  ->  %s"
              % (spread, STAGGER_UNIT_US, breadth, later, "BREADTH-FIRST" if ok else "NOT breadth-first"))
    print("verdict: %d of %d launches placed breadth-first" % (sum(verdicts), len(verdicts)))
    return 0 if all(verdicts) else 1


if __name__ == "__main__":
    sys.exit(main())
