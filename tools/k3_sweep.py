"""Times the fused loss kernel alone (events around the C-ABI call) over S, B, H and VEC."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svbrdf_estimation_amd import _native, environment

def maps(B, H, gen, tied=True):
    n = torch.randn(B, 3, H, H, generator=gen) * 0.3
    n[:, 2] = 1 + n[:, 2].abs(); n = n / n.norm(dim=1, keepdim=True)
    r = torch.rand(B, 1, H, H, generator=gen).expand(B, 3, H, H) if tied else torch.rand(B, 3, H, H, generator=gen)
    return torch.cat((n, torch.rand(B, 3, H, H, generator=gen), r, torch.rand(B, 3, H, H, generator=gen)), 1).contiguous()

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2] * 1e3

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1)
    if os.environ.get("SWEEP") == "short":
        cfgs = [(8, 256, S, True, g) for S in (1, 4, 9, 18) for g in (True, False)]
    else:
        cfgs = [(8, 256, S, tied, g) for S in (1, 2, 4, 9, 18) for tied in (True, False) for g in (True, False)]
        cfgs += [(32, 256, 9, True, True), (2, 256, 9, True, True), (8, 512, 9, True, True), (8, 512, 32, True, True)]
    for (B, H, S, tied, grad) in cfgs:
        inp, tgt = maps(B, H, gen, tied).to(dev), maps(B, H, gen, tied).to(dev)
        torch.manual_seed(0)
        table = environment.BatchSceneSampler(B, S // 3, S - S // 3).sample()
        # the step path hands tables of <= 96 rows over by value (k_rendering_loss_inl); SWEEP_DEVICE_TABLE=1 times
        # the device-pointer kernel instead
        if os.environ.get("SWEEP_DEVICE_TABLE") or B * S > _native.host_scenes_max_rows():
            table = table.to(dev)
        us = timeit(lambda: _native.rendering_loss(inp, tgt, table, 0.1, want_grad=grad))
        px = B * H * H
        print("B=%-3d H=%-4d S=%-3d tied=%d grad=%d  %8.1f us  %7.2f ns/pixel  %6.3f ns/pixel-scene  alg %.0f GB/s" % (
            B, H, S, tied, grad, us, us * 1e3 / px, us * 1e3 / px / S, (144 if grad else 96) * px / us / 1e3), flush=True)

