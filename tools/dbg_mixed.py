#!/usr/bin/env python3
"""debug: step time of the module path for B=8 RenderingLoss and B=16 MixedLoss, one stream, fast backward on/off"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synthetic_maps
from svbrdf_estimation_amd import losses, renderers
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
g = torch.Generator().manual_seed(13)
for name, B, fn in (("render B8", 8, losses.RenderingLoss(renderers.LocalRenderer())),
                    ("mixed B8", 8, losses.MixedLoss(renderers.LocalRenderer())),
                    ("mixed B16", 16, losses.MixedLoss(renderers.LocalRenderer())),
                    ("render B16", 16, losses.RenderingLoss(renderers.LocalRenderer()))):
    sets = [(synthetic_maps(g, B, 256).to(dev).requires_grad_(True), synthetic_maps(g, B, 256).to(dev)) for _ in range(4)]
    for fast in (True, False):
        losses._FAST_BACKWARD = fast
        for stream in (None, torch.cuda.Stream(dev)):
            if stream is not None:
                torch.cuda.set_stream(stream)
            def run(n):
                for k in range(n):
                    a, t = sets[k % 4]
                    a.grad = None
                    fn(a, t).backward()
            run(60); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(300); host = time.perf_counter() - t0; torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            print("%-10s fast_backward=%d stream=%s: %.1f us/step wall, %.1f us host" % (
                name, fast, "default" if stream is None else "side", 1e6 * wall / 300, 1e6 * host / 300), flush=True)
            torch.cuda.set_stream(torch.cuda.default_stream(dev))
