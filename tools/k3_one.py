"""Launches the fused loss kernel of config 2 a few times (for rocprofv3 --pmc passes over one kernel build:
SVBRDF_HIP_LIB=<lib> rocprofv3 --kernel-trace --pmc ... -- python3 tools/k3_one.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SVBRDF_NO_HOST_EXT", "1")
import torch
from svbrdf_estimation_amd import _native, environment
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k3_sweep import maps

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
B, H, S = 8, 256, int(os.environ.get("K3_S", "9"))
inp, tgt = maps(B, H, gen).to(dev), maps(B, H, gen).to(dev)
torch.manual_seed(0)
table = environment.BatchSceneSampler(B, S // 3, S - S // 3).sample().to(dev)
for _ in range(int(os.environ.get("K3_N", "12"))):
    _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True)
torch.cuda.synchronize()
