"""INTEGRATION.md section 1 executed on the GPU: ``svbrdf_estimation_amd.install()`` patches flat modules named like the
reference's (``renderers``, ``losses``), and a stand-in of the reference's training script -- its import lines
(development/multiImage_pytorch/main.py:8,12), its loss construction (:82-89) and its step (:113-118) -- then runs on
the fused kernel.  The real reference does not exist on the GPU box, so the two modules are minimal stand-ins written
here (classes with the reference's names whose methods raise); the CPU suite runs the same patch against the real
reference (tests/test_host_logic.py::test_install_patches_the_reference_modules_in_a_fresh_interpreter)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STANDIN_RENDERERS = '''
class LocalRenderer:
    def render(self, scene, svbrdf):
        raise AssertionError("the stand-in's eager renderer was called: install() did not take")
class RednerRenderer:
    pass
'''
_STANDIN_LOSSES = '''
import torch.nn as nn
class SVBRDFL1Loss(nn.Module):
    def forward(self, input, target):
        raise AssertionError("stand-in loss called")
class RenderingLoss(nn.Module):
    def __init__(self, renderer):
        super().__init__()
    def forward(self, input, target):
        raise AssertionError("stand-in loss called")
class MixedLoss(nn.Module):
    def __init__(self, renderer, l1_weight=0.1):
        super().__init__()
    def forward(self, input, target):
        raise AssertionError("stand-in loss called")
'''
_STANDIN_MAIN = '''
import sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
import svbrdf_estimation_amd
svbrdf_estimation_amd.install()                       # the one added line (INTEGRATION.md section 1)

import torch
from losses import MixedLoss                          # main.py:8
from renderers import LocalRenderer, RednerRenderer   # main.py:12
import synth

device = torch.device("cuda:0")
B, H = 2, 32
host_svbrdf = torch.from_numpy(synth.make_maps(7, B, H))

# the dataloader's call (dataset.py:94-98, :206-212), main process, HOST tensors, default install(): the patched renderer
# serves it on the GPU and hands back a CPU tensor that cats with the photos read from disk
import renderers
from svbrdf_estimation_amd import environment as env
generated = []
for b in range(B):
    scene = env.Scene(env.Camera(torch.tensor([0.05, -0.1, 2.75])), env.Light(torch.tensor([0.3, 0.2, 2.197]), torch.tensor([30.0, 29.0, 31.0])))
    rendering = renderers.LocalRenderer().render(scene, host_svbrdf[b].unsqueeze(0))
    assert rendering.device.type == "cpu" and tuple(rendering.shape) == (1, 3, H, H)
    noise = torch.zeros_like(rendering).normal_(mean=0.0, std=0.005)
    rendering = torch.clamp(rendering + noise, min=0.0, max=1.0)
    generated.append(torch.cat([torch.zeros(0, 3, H, H), rendering], dim=0))
batch_inputs = torch.stack(generated)[:, 0].to(device)                    # main.py:107
batch_svbrdf = host_svbrdf.to(device)                                     # main.py:108


class Model(torch.nn.Module):                         # any network that ends in 12-channel maps (models.py:338-346)
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 9, 3, padding=1)

    def forward(self, x):
        from svbrdf_estimation_amd import losses as amd_losses
        return amd_losses.decode_head(torch.tanh(self.conv(x)))


torch.manual_seed(5)
model = Model().to(device)
optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)      # main.py:74
loss_renderer = LocalRenderer()                                 # main.py:82-86
loss_function = MixedLoss(loss_renderer)                        # main.py:89
assert loss_function.rendering_loss.uses_fused_kernel()
model.train()
seen = []
for i in range(12):                                             # main.py:104-118
    optimizer.zero_grad()
    outputs = model(batch_inputs)
    loss = loss_function(outputs, batch_svbrdf)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0 for p in model.parameters())
    optimizer.step()
    seen.append(loss.item())
# same seed, same step through the package's own names: the patched classes ARE the package's
torch.manual_seed(99)
a = loss_function(outputs.detach(), batch_svbrdf).item()
torch.manual_seed(99)
from svbrdf_estimation_amd import losses as L, renderers as R
b = L.MixedLoss(R.LocalRenderer())(outputs.detach(), batch_svbrdf).item()
assert a == b, (a, b)
print("STANDIN-MAIN-OK first %%.5f last %%.5f" %% (sum(seen[:3]) / 3, sum(seen[-3:]) / 3))
assert sum(seen[-3:]) < sum(seen[:3])
'''


def test_install_then_the_reference_shaped_training_script_runs_on_the_fused_kernel(tmp_path):
    (tmp_path / "renderers.py").write_text(_STANDIN_RENDERERS)
    (tmp_path / "losses.py").write_text(_STANDIN_LOSSES)
    (tmp_path / "main.py").write_text(_STANDIN_MAIN % {"root": ROOT, "tests": os.path.join(ROOT, "tests")})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "main.py"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "STANDIN-MAIN-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    print(r.stdout.strip().splitlines()[-1])


class _RenderingDataset:
    """a dataset whose items render a HOST map with the patched renderer, like SvbrdfDataset.__getitem__ (dataset.py:94-98)"""

    def __len__(self):
        return 2

    def __getitem__(self, i):
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import synth
        from svbrdf_estimation_amd import environment as env, renderers
        maps = torch.from_numpy(synth.make_maps(40 + i, 1, 16))
        scene = env.Scene(env.Camera(torch.tensor([0.0, 0.1, 2.75])), env.Light(torch.tensor([0.2, 0.1, 2.197]), torch.tensor([30.0, 30.0, 30.0])))
        return renderers.LocalRenderer().render(scene, maps)[0]


def test_dataloader_workers_forked_after_gpu_init_are_refused_spawned_ones_render():
    """INTEGRATION.md section 1, the table of what still raises: a FORKED DataLoader worker whose parent has initialised the
    GPU runtime cannot use the GPU -- the renderer says so (and names the ways out) instead of crashing in the runtime;
    num_workers=0 (the reference's main.py:63) renders in the main process; a SPAWNED worker has a runtime of its own and
    renders."""
    import torch
    from svbrdf_estimation_amd import NativeLibraryError
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda:0")                                         # the parent has initialised the runtime
    main_process = torch.stack(list(torch.utils.data.DataLoader(_RenderingDataset(), batch_size=None, num_workers=0)))
    assert main_process.device.type == "cpu" and tuple(main_process.shape) == (2, 3, 16, 16)
    with pytest.raises(NativeLibraryError, match="forked worker"):
        list(torch.utils.data.DataLoader(_RenderingDataset(), batch_size=None, num_workers=1, multiprocessing_context="fork"))
    spawned = torch.stack(list(torch.utils.data.DataLoader(_RenderingDataset(), batch_size=None, num_workers=1,
                                                           multiprocessing_context="spawn", timeout=300)))
    assert torch.equal(spawned, main_process)
