#!/usr/bin/env python3
"""debug: why is the first default-stream leg of mixed B16 with the engine-free backward host-bound at ~250 us/step?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synthetic_maps
from svbrdf_estimation_amd import losses, renderers
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
g = torch.Generator().manual_seed(13)
order = sys.argv[1] if len(sys.argv) > 1 else "default-first"
fn = losses.MixedLoss(renderers.LocalRenderer())
if "pre8" in sys.argv:
    fn8 = losses.RenderingLoss(renderers.LocalRenderer()) if "render8" in sys.argv else fn
    s8 = [(synthetic_maps(g, 8, 256).to(dev).requires_grad_(True), synthetic_maps(g, 8, 256).to(dev)) for _ in range(4)]
    for k in range(400):
        a, t = s8[k % 4]
        a.grad = None
        fn8(a, t).backward()
    torch.cuda.synchronize()
    if "free8" in sys.argv:
        del s8
    print("ran 400 steps at B=8 first (%s)" % type(fn8).__name__, flush=True)
B = 16
sets = [(synthetic_maps(g, B, 256).to(dev).requires_grad_(True), synthetic_maps(g, B, 256).to(dev)) for _ in range(4)]
side = torch.cuda.Stream(dev)
legs = [("default", None), ("side", side), ("default", None)] if order == "default-first" else [("side", side), ("default", None), ("side", side)]
for name, st in legs:
    torch.cuda.set_stream(st if st is not None else torch.cuda.default_stream(dev))
    for rep in range(2):
        ms0 = torch.cuda.memory_stats(dev)
        tf = tb = 0.0
        def run(n, rec):
            global tf, tb
            for k in range(n):
                a, t = sets[k % 4]
                a.grad = None
                t0 = time.perf_counter(); loss = fn(a, t); t1 = time.perf_counter(); loss.backward(); t2 = time.perf_counter()
                if rec:
                    tf += t1 - t0; tb += t2 - t1
        run(60, False); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(300, True); host = time.perf_counter() - t0; torch.cuda.synchronize(); wall = time.perf_counter() - t0
        ms1 = torch.cuda.memory_stats(dev)
        if rep == 0 and name == legs[0][0] and "trace" in sys.argv:
            per = []
            for k in range(40):
                a, t = sets[k % 4]
                a.grad = None
                t0 = time.perf_counter(); fn(a, t).backward(); per.append(1e6 * (time.perf_counter() - t0))
            torch.cuda.synchronize()
            print("  per-step host us:", " ".join("%.0f" % v for v in per), flush=True)
        print("%s leg rep %d: wall %.1f us/step host %.1f (forward call %.1f, backward call %.1f); device mallocs %d frees %d, alloc retries %d, type(loss)=%s" % (
            name, rep, 1e6 * wall / 300, 1e6 * host / 300, 1e6 * tf / 300, 1e6 * tb / 300,
            ms1["num_device_alloc"] - ms0["num_device_alloc"], ms1["num_device_free"] - ms0["num_device_free"],
            ms1["num_alloc_retries"] - ms0["num_alloc_retries"], type(fn(sets[0][0], sets[0][1])).__name__), flush=True)
torch.cuda.set_stream(torch.cuda.default_stream(dev))
