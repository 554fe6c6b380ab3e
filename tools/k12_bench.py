"""Times K1 (render forward) and K2 (render backward) alone; prints algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from svbrdf_estimation_amd import _native, environment
from k3_sweep import maps, timeit

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1)
    # B=288/576, S=1: 0.9-1.8 GB of maps per launch, far beyond the 256 MiB Infinity Cache -> genuine HBM rates
    for (B, H, S, tied) in [(8, 256, 9, True), (8, 256, 9, False), (8, 256, 1, True), (72, 256, 1, True),
                            (288, 256, 1, True), (576, 256, 1, False), (32, 256, 9, True), (8, 512, 32, True)]:
        m = maps(B, H, gen, tied).to(dev)
        torch.manual_seed(0)
        table = environment.BatchSceneSampler(B, S // 3, S - S // 3).sample().to(dev)
        go = torch.randn(B, S, 3, H, H, device=dev)
        px = B * H * H
        for vec in ("1", "2", "4"):
            os.environ["SVBRDF_K1_VEC"] = vec
            os.environ["SVBRDF_K2_VEC"] = vec
            t1 = timeit(lambda: _native.render_fwd(m, table))
            t2 = timeit(lambda: _native.render_bwd(m, table, go))
            b1 = (12 + 3 * S) * 4 * px
            b2 = (24 + 3 * S) * 4 * px
            print("B=%-3d H=%-4d S=%-3d tied=%d vec=%s | K1 %7.1f us %6.0f GB/s (%.2f of 8TB/s) | K2 %7.1f us %6.0f GB/s (%.2f)" % (
                B, H, S, tied, vec, t1, b1 / t1 / 1e3, b1 / t1 / 1e3 / 8000, t2, b2 / t2 / 1e3, b2 / t2 / 1e3 / 8000), flush=True)
