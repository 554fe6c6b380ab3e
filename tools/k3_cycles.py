"""Per-wave cycle counts of K3's scene loop (needs a -DSVBRDF_TIMING=1 build: SVBRDF_HIP_LIB=...).
Prints, for several batch sizes (= waves per SIMD while one round fits), cycles per scene iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SVBRDF_NO_HOST_EXT", "1")
import torch
from svbrdf_estimation_amd import _native, environment
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k3_sweep import maps

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
H = 256
for S in (9, 18):
    for B in (1, 2, 3, 4, 8):
        inp, tgt = maps(B, H, gen).to(dev), maps(B, H, gen).to(dev)
        torch.manual_seed(0)
        table = environment.BatchSceneSampler(B, S // 3, S - S // 3).sample().to(dev)
        for _ in range(3):
            loss, g = _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True)
        torch.cuda.synchronize()
        cyc, ticks = g[:, 0].flatten().double(), g[:, 1].flatten().double()
        wg = B * H * H // 256
        print("S=%-2d B=%d  workgroups %4d (%.1f per CU)  loop cycles/scene: mean %.0f  min %.0f  max %.0f   "
              "ns/scene (100 MHz ticks): %.0f   => clock %.2f GHz" % (
                  S, B, wg, wg / 256.0, cyc.mean().item() / S, cyc.min().item() / S, cyc.max().item() / S,
                  ticks.mean().item() * 10 / S, cyc.mean().item() / (ticks.mean().item() * 10)), flush=True)
