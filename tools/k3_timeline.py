"""Occupancy over time of one K3 launch (config 2) from per-wave stamps of a -DSVBRDF_TIMING=1 build
(SVBRDF_HIP_LIB=<that build>: `bash tools/build_variant.sh tim -DSVBRDF_TIMING=1`; the scene-split layouts this tool also
read in round 4 left the source in round 5).  Prints waves resident / inside the
scene loop per SIMD in 24 time bins, and where the first and last microseconds of the launch go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SVBRDF_NO_HOST_EXT", "1")
import numpy as np, torch
from svbrdf_estimation_amd import _native, environment
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import synthetic_maps


def maps(B, H, gen, tied=True):
    return synthetic_maps(gen, B, H, tied=tied)

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
B, H, S = int(os.environ.get("K3_B", "8")), 256, int(os.environ.get("K3_S", "9"))
G = 1      # waves sharing a pixel's renders (the scene-split experiment of round 4: always 1 now)
inp, tgt = maps(B, H, gen).to(dev), maps(B, H, gen).to(dev)
torch.manual_seed(0)
table = environment.BatchSceneSampler(B, S // 3, S - S // 3).sample()
for _ in range(3):
    loss, g = _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True)
torch.cuda.synchronize()
TICK = 0.01                                                             # us per s_memrealtime tick (100 MHz)
M = 1 << 24


def lane0(plane_index):
    return g[:, plane_index].flatten().cpu().numpy().astype(np.int64)[::64]


if G == 1:
    dur, start = lane0(1), lane0(2)
    entry, exit_ = lane0(3), lane0(4)
else:                                                                   # planes 3 w .. 3 w + 2 = wave w of the workgroup
    dur = np.concatenate([lane0(3 * w + 1) for w in range(G)])
    start = np.concatenate([lane0(3 * w + 2) for w in range(G)])
    entry = exit_ = None
t0 = start.min() if entry is None else entry.min()
start = (start - t0) % M
end = start + dur
T = end.max()
if entry is not None:
    entry, exit_ = (entry - t0) % M, (exit_ - t0) % M
    T = max(T, exit_.max())
nsimd = 1024.0
print("layout: %s   waves %d   loop start: median %.2f us, max %.2f us   last loop end %.2f us   mean loop %.2f us" % (
    "256 pixels x all renders" if G == 1 else "64 pixels x %d waves" % G, len(dur), np.median(start) * TICK,
    start.max() * TICK, end.max() * TICK, dur.mean() * TICK))
if entry is not None:
    print("first wave entry -> its loop start %.2f us;   entry of the last first-round wave %.2f us;   last exit %.2f us" % (
        (start[np.argmin(entry)] - entry.min()) * TICK, np.sort(entry)[min(len(entry), 4096) - 1] * TICK, exit_.max() * TICK))
    print("entry -> loop start (plane loads, prepare, first geometry): median %.2f us, first round %.2f us, second round %.2f us" % (
        np.median(start - entry) * TICK, np.median((start - entry)[np.argsort(entry)[:4096]]) * TICK,
        np.median((start - entry)[np.argsort(entry)[4096:]]) * TICK if len(entry) > 4096 else float("nan")))
edges = np.linspace(0, T, 25)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    active = int(((start <= mid) & (end > mid)).sum())
    res = "" if entry is None else "resident %.2f  " % (((entry <= mid) & (exit_ > mid)).sum() / nsimd)
    print("t = %5.1f us   %sin the scene loop %.2f waves per SIMD  %s" % (mid * TICK, res, active / nsimd, "#" * (active // 128)))
# issue-weighted loss: a SIMD issues at its peak with >= 2 waves in the loop; below that it idles part of the time
for lo, hi, name in ((0, 4 / TICK, "first 4 us"), (T - 10 / TICK, T, "last 10 us")):
    ts = np.linspace(max(lo, 0), hi, 200)
    occ = np.array([((start <= t) & (end > t)).sum() / nsimd for t in ts])
    print("%s: mean %.2f waves per SIMD in the loop" % (name, occ.mean()))
if G == 1:
    # the gap between two launches back to back on one stream: last exit stamp of the first, first entry stamp of the second
    gaps = []
    for _ in range(5):
        torch.cuda.synchronize()
        _, g1 = _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True)
        _, g2 = _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True)
        torch.cuda.synchronize()
        e1 = g1[:, 3].flatten().cpu().numpy().astype(np.int64)[::64]
        x1 = g1[:, 4].flatten().cpu().numpy().astype(np.int64)[::64]
        e2 = g2[:, 3].flatten().cpu().numpy().astype(np.int64)[::64]
        x2 = g2[:, 4].flatten().cpu().numpy().astype(np.int64)[::64]
        base = e1.min()
        gaps.append((((e2 - base) % M).min() - ((x1 - base) % M).max(), ((x1 - base) % M).max(), ((x2 - base) % M).max()))
    print("two launches back to back: first entry -> last exit of launch 1 / gap to the first entry of launch 2 / last exit of launch 2 (us):")
    for gap, x1m, x2m in gaps:
        print("   %.2f / %.2f / %.2f" % (x1m * TICK, gap * TICK, x2m * TICK))
