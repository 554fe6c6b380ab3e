"""Times every K3 variant at config 2 (rendering loss, mixed loss, head-fused; with / without gradient)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svbrdf_estimation_amd import _native, environment
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k3_sweep import maps, timeit

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
B, H, S = 8, 256, 9
inp, tgt = maps(B, H, gen).to(dev), maps(B, H, gen).to(dev)
enc = (torch.rand(B, 9, H, H, generator=gen) * 2 - 1).to(dev)
torch.manual_seed(0)
table = environment.BatchSceneSampler(B, 3, 6).sample()
for head in (False, True):
    for l1w in (0.0, 0.1):
        for grad in (True, False):
            x = enc if head else inp
            us = timeit(lambda: _native.rendering_loss(x, tgt, table, 0.1, want_grad=grad, l1_weight=l1w, head=head))
            print("head=%d l1=%d grad=%d  %6.1f us" % (head, l1w != 0, grad, us), flush=True)
