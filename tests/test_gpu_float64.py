"""float64 maps through the plugin interface (SURVEY section 8 row b: `render` / `RenderingLoss.forward` are dtype-agnostic in
the reference, renderers.py:67-104, losses.py:29-52).  With double maps the reference computes in MIXED precision -- pixel
grid, positions and colours are float32 (torch.linspace's default dtype, torch.Tensor(...)), what touches the maps is
promoted to double -- and so does the engine: float32 exact-rounded geometry (the same code as the float32 path), double
shading (svbrdf_render_{fwd,bwd}_f64), the losses composed from the renders through autograd.

Fixture g15_float64.npz: the reference run on double maps (tests/golden/make_golden.py g15_float64).  Tolerances: the two
float32 geometry factors the engine evaluates with 1-ULP primitives instead of the reference's op sequence ((1-VH)^5 and
colour/|L|^2; wo.h as FMAs) differ by ~1e-7 relative, and with them the double results: 2e-6 relative + 2e-7 of max for
renderings, 2e-6 for losses, 1e-5 + 1e-6 of max for gradients."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _close(a, b, rtol, afrac, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    tol = rtol * np.abs(b) + afrac * np.abs(b).max()
    bad = np.abs(a - b) > tol
    assert not bad.any(), "%s: %d/%d outside %.0e rel + %.0e*max (worst %.3e of max)" % (
        what, bad.sum(), b.size, rtol, afrac, np.abs(a - b).max() / np.abs(b).max())


def test_float64_render_forward_and_backward_match_the_reference(dev, golden):
    from svbrdf_estimation_amd import environment as env, renderers
    g = golden("g15_float64.npz")
    R = renderers.LocalRenderer()
    x = torch.from_numpy(g["render_maps"]).to(dev).requires_grad_(True)
    assert x.dtype == torch.float64
    table = g["render_scenes"]
    scenes = env.scenes_from_table(torch.from_numpy(table))
    rend = torch.stack([R.render(sc, x) for sc in scenes], dim=1)                # the reference-shaped calls, one per scene
    assert rend.dtype == torch.float64 and tuple(rend.shape) == tuple(g["render_out"].shape)
    _close(rend.detach().cpu().numpy(), g["render_out"], 2e-6, 2e-7, "float64 render")
    (rend * torch.from_numpy(g["render_cot"]).to(dev)).sum().backward()
    grad = x.grad.cpu().numpy()
    _close(grad, g["render_grad"], 1e-5, 1e-6, "float64 render gradient")
    assert not grad[0, 6:9, :2].any()                                            # roughness below the clamp: exactly zero
    # all scenes in one launch (render_many), host and device tables, a non-contiguous view of the maps
    many = R.render_many(torch.from_numpy(table), x.detach())
    assert torch.equal(many, rend.detach())
    dev_table = torch.from_numpy(table).to(dev).unsqueeze(0).expand(2, -1, -1).contiguous()
    assert torch.equal(R.render_many(dev_table, x.detach()), many)
    wide = torch.zeros(2, 12, 16, 32, device=dev, dtype=torch.float64)
    wide[..., ::2] = x.detach()
    assert torch.equal(R.render(scenes[0], wide[..., ::2]), rend[:, 0].detach())
    # a float32 copy of the same maps through the float32 kernels: same values to float32 accuracy (different inputs by 2^-30)
    r32 = R.render_many(torch.from_numpy(table), x.detach().float())
    assert r32.dtype == torch.float32
    _close(r32.cpu().numpy(), g["render_out"], 1e-4, 1e-5, "float32 render of the rounded maps")


def test_float64_losses_match_the_reference(dev, golden):
    """RenderingLoss and MixedLoss on double inputs: the scenes are re-drawn from the seed (the sampler is bit-exact), the
    value and the gradient compared with the reference's autograd"""
    from svbrdf_estimation_amd import losses, renderers
    g = golden("g15_float64.npz")
    tgt = torch.from_numpy(g["loss_target"]).to(dev)
    for name, fn in (("loss", losses.RenderingLoss(renderers.LocalRenderer())), ("mixed", losses.MixedLoss(renderers.LocalRenderer()))):
        x = torch.from_numpy(g["loss_input"]).to(dev).requires_grad_(True)
        rl = fn if name == "loss" else fn.rendering_loss
        torch.manual_seed(int(g["loss_rng_seed"]))
        # same draws as the reference's forward (to the last bit on the host the fixture was made on; torch's CPU sin / cos /
        # exp differ in the last bit between hosts), then the recorded table itself for the comparison of values
        np.testing.assert_allclose(rl.sample_scene_table(2).numpy(), g[name + "_scenes"], rtol=1e-6, atol=2e-7)
        rl.sample_scene_table = lambda B, _t=torch.from_numpy(g[name + "_scenes"]): _t.clone()
        val = fn(x, tgt)
        assert val.dtype == torch.float64 and val.dim() == 0
        assert abs(val.item() - float(g[name + "_value"])) <= 2e-6 * abs(float(g[name + "_value"])), (name, val.item())
        val.backward()
        _close(x.grad.cpu().numpy(), g[name + "_grad"], 1e-5, 1e-6, "float64 %s gradient" % name)
    # a float32 target with a double input is promoted, as torch promotes in the reference
    torch.manual_seed(3)
    assert losses.RenderingLoss(renderers.LocalRenderer())(torch.from_numpy(g["loss_input"]).to(dev), tgt.float()).dtype == torch.float64
    # ... and the other way round: float32 input, double target (this fell into the float32 kernel and raised)
    torch.manual_seed(3)
    x32 = torch.from_numpy(g["loss_input"]).float().to(dev).requires_grad_(True)
    v = losses.RenderingLoss(renderers.LocalRenderer())(x32, tgt)
    assert v.dtype == torch.float64
    v.backward()
    assert x32.grad.dtype == torch.float32 and torch.isfinite(x32.grad).all()
    # MixedLoss on double maps honours the L1 loss's epsilon like the float32 fused path (it was dropped: always 0.01)
    vals = []
    from svbrdf_estimation_amd import _hostext
    _hostext.set_enabled(False)         # the ctypes host path draws its scenes through sample_scene_table (patched below)
    for e in (0.01, 0.05):
        mixed = losses.MixedLoss(renderers.LocalRenderer())
        mixed.l1_loss.epsilon_l1 = e
        mixed.rendering_loss.sample_scene_table = lambda B, _t=torch.from_numpy(g["mixed_scenes"]): _t.clone()
        vals.append((mixed(torch.from_numpy(g["loss_input"]).to(dev), tgt).item(),
                     mixed(torch.from_numpy(g["loss_input"]).float().to(dev), tgt.float()).item()))
    _hostext.set_enabled(True)
    assert abs(vals[0][0] - vals[1][0]) > 1e-4 * abs(vals[0][0])                      # the epsilon matters ...
    assert all(abs(d - f) <= 2e-5 * abs(d) for d, f in vals), vals                    # ... and both precisions agree on it


def test_float64_gradcheck_of_the_analytic_adjoint(dev):
    """torch.autograd.gradcheck (central differences of the forward kernel in double against the analytic backward
    kernel): an independent check of the adjoint's algebra -- the oracle's adjoint is the same derivation, finite
    differences are not.  Smooth region only (no clamp is active: gradcheck cannot cross a kink)."""
    from svbrdf_estimation_amd import environment as env, renderers
    R = renderers.LocalRenderer()
    gen = torch.Generator().manual_seed(5)
    H = 4
    n = torch.randn(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.15
    n[:, 2] = 1.0
    n = n / n.norm(dim=1, keepdim=True)
    d = torch.rand(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.8 + 0.1
    r = torch.rand(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.6 + 0.2
    s = torch.rand(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.8 + 0.1
    maps = torch.cat((n, d, r, s), dim=1).to(dev).requires_grad_(True)
    for cam, light in (([0.2, -0.3, 2.0], [0.5, 0.4, 1.5]), ([-0.8, 0.6, 1.2], [0.9, -0.7, 0.8])):
        scene = env.Scene(env.Camera(cam), env.Light(light, [20.0, 30.0, 40.0]))
        assert torch.autograd.gradcheck(lambda m: R.render(scene, m), (maps,), eps=1e-6, atol=1e-7, rtol=1e-5,
                                        nondet_tol=0.0, fast_mode=False)
