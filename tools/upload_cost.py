"""What does the per-step scene-table upload cost once the loop is GPU-bound?  (run on the GPU box)
Times the bench step (sample on host -> pinned ring -> H2D -> K3 -> autograd) against the same step
with a device-resident table (no sampling, no upload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svbrdf_estimation_amd import _hostext, losses, renderers
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k3_sweep import maps  # noqa

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
inp = maps(8, 256, gen).to(dev).requires_grad_(True)
tgt = maps(8, 256, gen).to(dev)
fn = losses.RenderingLoss(renderers.LocalRenderer())
ext = _hostext.module()
assert ext is not None
torch.autograd.set_multithreading_enabled(False)
td = fn.sample_scene_table(8).to(dev)
stream = torch.cuda.current_stream(dev).cuda_stream

def step_upload():
    inp.grad = None
    fn(inp, tgt).backward()

def step_resident():
    inp.grad = None
    ext.fused_loss_with_scenes(inp, tgt, td, 0.1, 0.0, 0.01, stream, False).backward()

def T(f, n=3000):
    for _ in range(500): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    h = time.perf_counter() - t
    torch.cuda.synchronize()
    return h / n * 1e6, (time.perf_counter() - t) / n * 1e6

for rep in range(3):
    print("upload per step : host %.1f us  wall %.1f us" % T(step_upload))
    print("resident table  : host %.1f us  wall %.1f us" % T(step_resident))
