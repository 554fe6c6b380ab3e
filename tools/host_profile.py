"""Where does the host time of one bench step go?  (run on the GPU box)"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svbrdf_estimation_amd import _native, losses, renderers
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k3_sweep import maps  # noqa

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
inp = maps(8, 256, gen).to(dev).requires_grad_(True)
tgt = maps(8, 256, gen).to(dev)
fn = losses.RenderingLoss(renderers.LocalRenderer())

def step():
    inp.grad = None
    loss = fn(inp, tgt)
    loss.backward()

def T(f, n=300):
    for _ in range(30): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    h = time.perf_counter() - t
    torch.cuda.synchronize()
    return h / n * 1e6, (time.perf_counter() - t) / n * 1e6

print("step: host %.1f us, wall %.1f us" % T(step))
table = fn.sample_scene_table(8)
print("sample_scene_table: %.1f us" % T(lambda: fn.sample_scene_table(8))[0])
print("table.to(dev): %.1f us" % T(lambda: table.to(dev, non_blocking=True))[0])
td = table.to(dev)
print("native.rendering_loss: host %.1f us wall %.1f" % T(lambda: _native.rendering_loss(inp.detach(), tgt, td)))
def fb():
    inp.grad = None
    l = losses._FusedRenderingLoss.apply(inp, tgt, td, 0.1); l.backward()
print("Function fwd+bwd: host %.1f us wall %.1f" % T(fb))
print("torch.empty_like: %.1f us" % T(lambda: torch.empty_like(tgt))[0])
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])

# ---- finer split of the autograd cost
import time
def split():
    fw = bw = 0.0
    for _ in range(300):
        inp.grad = None
        t0 = time.perf_counter()
        l = losses._FusedRenderingLoss.apply(inp, tgt, td, 0.1)
        t1 = time.perf_counter()
        l.backward()
        t2 = time.perf_counter()
        fw += t1 - t0; bw += t2 - t1
    print("Function.apply (forward): %.1f us   loss.backward(): %.1f us" % (fw / 300 * 1e6, bw / 300 * 1e6))
split()
x = torch.ones(1, device=dev, requires_grad=True)
def tiny():
    y = (x * 2.0).sum(); y.backward()
print("tiny builtin-op graph fwd+bwd: host %.1f us" % T(tiny)[0])
class Id(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a.view(())
    @staticmethod
    def backward(ctx, g): return g.reshape(1)
def tiny_fn():
    y = Id.apply(x); y.backward()
print("trivial python Function fwd+bwd: host %.1f us" % T(tiny_fn)[0])

# ---- native host path (C++ extension)
from svbrdf_estimation_amd import _hostext
ext = _hostext.module()
print("host extension:", ext)
if ext is not None:
    print("C++ sampler: %.1f us" % T(lambda: ext.sample_scene_table(8, 3, 6))[0])
    print("step via RenderingLoss (ext path): host %.1f us wall %.1f" % T(step))
    os.environ["X"] = "1"

if ext is not None:
    def split_ext():
        fw = bw = oth = 0.0
        raw = _native._raw_stream(dev)
        for _ in range(300):
            t0 = time.perf_counter()
            inp.grad = None
            t1 = time.perf_counter()
            l = ext.fused_loss(inp, tgt, 3, 6, 0.1, 0.0, 0.01, raw, False)
            t2 = time.perf_counter()
            l.backward()
            t3 = time.perf_counter()
            oth += t1 - t0; fw += t2 - t1; bw += t3 - t2
        torch.cuda.synchronize()
        print("ext path: grad=None %.1f us, fused_loss() %.1f us (C++ sampler %.1f), backward() %.1f us" % (
            oth / 300 * 1e6, fw / 300 * 1e6, T(lambda: ext.sample_scene_table(8, 3, 6))[0], bw / 300 * 1e6))
    split_ext()
    tds = td
    def fwd_only():
        with torch.no_grad():
            ext.fused_loss_with_scenes(inp.detach(), tgt, tds, 0.1, 0.0, 0.01, _native._raw_stream(dev), False)
    print("fused_loss_with_scenes, no grad, no sampling: host %.1f us wall %.1f" % T(fwd_only))
