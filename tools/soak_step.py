#!/usr/bin/env python3
"""Soak of the training-loop step (forward + loss.backward() through the engine entered from the extension): N steps on
small maps, host RSS / device memory / allocator counters before and after -- a leak in the host path (graph tasks, nodes,
cached tensors) would show as growth per step.  Also through a non-leaf input and with MixedLoss.
    python tools/soak_step.py [steps]"""
import os
import resource
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                        # noqa: E402

from bench import synthetic_maps                    # noqa: E402
from svbrdf_estimation_amd import _native, losses, renderers, synthesis   # noqa: E402


def rss_mib():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * resource.getpagesize() / 2 ** 20


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    dev = torch.device("cuda:0")
    torch.autograd.set_multithreading_enabled(False)
    gen = torch.Generator().manual_seed(1)
    x = synthetic_maps(gen, 8, 32).to(dev).requires_grad_(True)
    t = synthetic_maps(gen, 8, 32).to(dev)
    w = torch.ones(1, device=dev, requires_grad=True)
    cases = (("RenderingLoss, leaf input", losses.RenderingLoss(renderers.LocalRenderer()), lambda: x),
             ("MixedLoss, non-leaf input", losses.MixedLoss(renderers.LocalRenderer()), lambda: x * w),
             ("render_inputs (one launch per call)", None, None))
    for name, fn, inp in cases:
        def step():
            if fn is None:
                synthesis.render_inputs(t, 2)
                return
            x.grad = None
            w.grad = None
            fn(inp(), t).backward()
        for _ in range(2000):
            step()
        torch.cuda.synchronize()
        r0, m0, l0 = rss_mib(), torch.cuda.memory_allocated(dev), _native.launch_count()
        for k in range(n):
            step()
            if k % 20000 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        r1, m1, l1 = rss_mib(), torch.cuda.memory_allocated(dev), _native.launch_count()
        print("%-38s %d steps: host RSS %.1f -> %.1f MiB (%+.1f bytes per step), device %d -> %d bytes, launches per step %.3f"
              % (name, n, r0, r1, (r1 - r0) * 2 ** 20 / n, m0, m1, (l1 - l0) / n), flush=True)
        assert (r1 - r0) * 2 ** 20 / n < 16.0, "host memory grows per step"
        assert abs(m1 - m0) <= 64 << 20


if __name__ == "__main__":
    main()
