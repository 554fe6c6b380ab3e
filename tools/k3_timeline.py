"""Occupancy over time of one K3 launch (config 2) from per-wave start stamps and loop durations
(-DSVBRDF_TIMING=1 build: SVBRDF_HIP_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SVBRDF_NO_HOST_EXT", "1")
import numpy as np, torch
from svbrdf_estimation_amd import _native, environment
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k3_sweep import maps

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
B, H, S = 8, 256, int(os.environ.get("K3_S", "9"))
inp, tgt = maps(B, H, gen).to(dev), maps(B, H, gen).to(dev)
torch.manual_seed(0)
table = environment.BatchSceneSampler(B, S // 3, S - S // 3).sample()
for _ in range(3):
    loss, g = _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True)
torch.cuda.synchronize()
TICK = 0.01                                                             # us per s_memrealtime tick (100 MHz)
dur = g[:, 1].flatten().cpu().numpy().astype(np.int64)[::64]           # one lane per wave, ticks
start = g[:, 2].flatten().cpu().numpy().astype(np.int64)[::64]
start = (start - start.min()) & 0xffffff
end = start + dur
T = end.max()
print("waves %d   loop start: median %.1f us, max %.1f us   last loop end %.1f us" % (
    len(dur), np.median(start) * TICK, start.max() * TICK, T * TICK))
edges = np.linspace(0, T, 25)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    active = int(((start <= mid) & (end > mid)).sum())
    print("t = %5.1f us   waves inside the scene loop: %5d  (%.2f per SIMD)  %s" % (mid * TICK, active, active / 1024.0, "#" * (active // 128)))
