"""K3 with many scenes per launch (long dispatches) -- for clock / PMC measurements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from svbrdf_estimation_amd import _native, environment
from k3_sweep import maps, timeit
if __name__ == "__main__":
    dev = torch.device("cuda:0"); gen = torch.Generator().manual_seed(1)
    B, H, S = 8, 256, int(os.environ.get("S", "288"))
    inp, tgt = maps(B, H, gen).to(dev), maps(B, H, gen).to(dev)
    torch.manual_seed(0)
    table = environment.BatchSceneSampler(B, S // 3, S - S // 3).sample().to(dev)
    us = timeit(lambda: _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True), n=20, warm=3)
    print("S=%d  %.1f us per launch, %.3f us per scene" % (S, us, us / S))
