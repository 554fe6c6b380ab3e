#!/usr/bin/env python3
"""200 reference-shaped `LocalRenderer().render(scene, svbrdf)` calls (forward, then forward + backward) for a
rocprofv3 --kernel-trace --memory-copy-trace run: the evidence that one call is ONE kernel dispatch with no H2D copy
command (the reference does three synchronous uploads per call, renderers.py:79,91,98).  Not product code.
    rocprofv3 --kernel-trace --memory-copy-trace --stats -d out -- python3 tools/render_call_trace.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import synthetic_maps  # noqa: E402
from svbrdf_estimation_amd import environment, renderers  # noqa: E402

dev = torch.device("cuda:0")
R = renderers.LocalRenderer()
scene = environment.Scene(environment.Camera([0.1, -0.2, 2.0]), environment.Light([0.4, 0.3, 1.5], [30.0, 30.0, 30.0]))
m = synthetic_maps(torch.Generator().manual_seed(1), 1, 256)[0].to(dev)
x = m.clone().requires_grad_(True)
cot = torch.randn(1, 3, 256, 256, device=dev)
R.render(scene, m)                                  # uploads xrow once (cached per width)
torch.cuda.synchronize()
print("MARK calls begin", flush=True)
t0 = time.perf_counter()
N = 200
for _ in range(N):
    R.render(scene, m)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(N):
    x.grad = None
    R.render(scene, x).backward(cot)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"calls_fwd": N, "calls_fwd_bwd": N, "us_per_fwd_call": 1e6 * (t1 - t0) / N,
                  "us_per_fwd_bwd_call": 1e6 * (t2 - t1) / N}))
