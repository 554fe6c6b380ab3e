"""``backward(create_graph=True)`` through the plugin interface (SURVEY section 8 row b).

The reference's ``LocalRenderer.render`` and its losses are compositions of differentiable torch ops (renderers.py:8-104,
losses.py:7-63), so gradients of gradients -- gradient penalties, Hessian-vector products -- work there.  The engine's
gradient comes out of hand-written kernels, which are constants to autograd; under ``create_graph=True`` its nodes switch
to the same loss composed from the float64 render kernels, whose backward has a forward-mode (dual number) companion
kernel, ``svbrdf_render_bwd_jvp_f64``.  float32 callers are promoted to double for that call.

Fixture g16_second_order.npz: the reference's own double backward (tests/golden/make_golden.py g16_second_order), float64.
Tolerances: first order as in test_gpu_float64.py (1e-5 relative + 1e-6 of max: the two float32 geometry factors the engine
evaluates with 1-ULP primitives differ by ~1e-7 relative from the reference's op sequence); second order the same class
of error once more through the chain rule: 5e-5 relative + 5e-6 of max; float32 callers 2e-4 + 2e-5 (their result is cast
back to float32 and the first-order factor in a penalty is the float32 one)."""
import ctypes
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


def _close_but(a, b, rtol, afrac, what, cap=8, loose=20.0):
    """_close, except that at most `cap` elements may sit within `loose` times the bound: the comparison partner here is
    the eager restatement on the CPU, whose float32 geometry goes through torch's MKL sqrt (1 ULP off on 0.7 % of values,
    tests/golden/make_golden.py) and the GGX denominator amplifies a last-bit difference of NH by 1e3-1e4 at highlight
    pixels -- the reference's own irreproducibility, the same one tests/tolerances.py counts for the first-order tests"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    tol = rtol * np.abs(b) + afrac * np.abs(b).max()
    err = np.abs(a - b)
    bad = err > tol
    assert bad.sum() <= cap and not (err > loose * tol).any(), "%s: %d/%d outside %.0e rel + %.0e*max (cap %d), worst %.3e of max" % (
        what, bad.sum(), b.size, rtol, afrac, cap, err.max() / np.abs(b).max())
    if bad.any():
        print("[second order] %s: %d element(s) of %d beyond the strict bound (highlight pixels), worst %.2e of max"
              % (what, bad.sum(), b.size, err.max() / np.abs(b).max()))


def test_render_second_order_matches_the_reference(dev, golden):
    """g = d<c, render(x)>/dx with create_graph, then d<g, v>/dx (Hessian-vector product) and d<g, v>/dc (= J v), all
    scenes in one launch and scene by scene through ``LocalRenderer.render`` as the reference's callers do"""
    from svbrdf_estimation_amd import environment as env, renderers
    g = golden("g16_second_order.npz")
    R = renderers.LocalRenderer()
    table = torch.from_numpy(g["render_scenes"])
    v = torch.from_numpy(g["render_v"]).to(dev)
    for how in ("render_many", "render"):
        x = torch.from_numpy(g["render_maps"]).to(dev).requires_grad_(True)
        cot = torch.from_numpy(g["render_cot"]).to(dev).requires_grad_(True)
        if how == "render_many":
            rend = R.render_many(table, x)
        else:
            rend = torch.stack([R.render(sc, x) for sc in env.scenes_from_table(table)], dim=1)
        (grad,) = torch.autograd.grad((rend * cot).sum(), x, create_graph=True)
        assert grad.requires_grad and grad.dtype == torch.float64
        _close(grad.detach().cpu().numpy(), g["render_grad"], 1e-5, 1e-6, how + ": gradient under create_graph")
        hv, jv = torch.autograd.grad((grad * v).sum(), (x, cot))
        _close(hv.cpu().numpy(), g["render_hvp"], 5e-5, 5e-6, how + ": Hessian-vector product")
        _close(jv.cpu().numpy(), g["render_jv"], 1e-5, 1e-6, how + ": J v")
        assert not hv[0, 6:9, :2].cpu().numpy().any()          # roughness below the clamp: no first, no second derivative


def test_gradgradcheck_of_the_dual_number_kernel(dev):
    """torch.autograd.gradgradcheck: central differences of the analytic backward kernel (K2, float64) against the
    dual-number kernel -- an independent check of the forward-mode algebra.  Smooth region only."""
    from svbrdf_estimation_amd import renderers
    gen = torch.Generator().manual_seed(9)
    H = 4
    n = torch.randn(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.15
    n[:, 2] = 1.0
    n = n / n.norm(dim=1, keepdim=True)
    d = torch.rand(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.8 + 0.1
    r = torch.rand(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.6 + 0.2
    s = torch.rand(1, 3, H, H, generator=gen, dtype=torch.float64) * 0.8 + 0.1
    maps = torch.cat((n, d, r, s), dim=1).to(dev).requires_grad_(True)
    table = torch.tensor([[0.2, -0.3, 2.0, 0.5, 0.4, 1.5, 20.0, 30.0, 40.0],
                          [-0.8, 0.6, 1.2, 0.9, -0.7, 0.8, 20.0, 30.0, 40.0]], dtype=torch.float32)
    R = renderers.LocalRenderer()
    assert torch.autograd.gradgradcheck(lambda m: R.render_many(table, m), (maps,), eps=1e-6, atol=1e-6, rtol=1e-4,
                                        nondet_tol=0.0, fast_mode=False)


def _second_order_of(fn, x, tgt, v):
    val = fn(x, tgt)
    (grad,) = torch.autograd.grad(val, x, create_graph=True)
    assert grad.requires_grad
    (pen,) = torch.autograd.grad((grad ** 2).sum(), x, retain_graph=True)
    (hv,) = torch.autograd.grad((grad * v).sum(), x)
    return val, grad.detach(), pen, hv


def test_float64_losses_second_order_match_the_reference(dev, golden):
    """RenderingLoss and MixedLoss on double inputs: gradient with create_graph, gradient of the penalty sum(g^2) and a
    Hessian-vector product against the reference's double backward (scenes: the recorded table)"""
    from svbrdf_estimation_amd import losses, renderers
    g = golden("g16_second_order.npz")
    tgt = torch.from_numpy(g["loss_target"]).to(dev)
    v = torch.from_numpy(g["loss_v"]).to(dev)
    for name, fn in (("loss", losses.RenderingLoss(renderers.LocalRenderer())), ("mixed", losses.MixedLoss(renderers.LocalRenderer()))):
        rl = fn if name == "loss" else fn.rendering_loss
        rl.sample_scene_table = lambda B, _t=torch.from_numpy(g[name + "_scenes"]): _t.clone()
        x = torch.from_numpy(g["loss_input"]).to(dev).requires_grad_(True)
        val, grad, pen, hv = _second_order_of(fn, x, tgt, v)
        assert abs(val.item() - float(g[name + "_value"])) <= 2e-6 * abs(float(g[name + "_value"]))
        _close(grad.cpu().numpy(), g[name + "_grad"], 1e-5, 1e-6, name + ": gradient under create_graph")
        _close(pen.cpu().numpy(), g[name + "_penalty_grad"], 5e-5, 5e-6, name + ": gradient of the gradient penalty")
        _close(hv.cpu().numpy(), g[name + "_hvp"], 5e-5, 5e-6, name + ": Hessian-vector product")


@pytest.mark.parametrize("host_path", ["native", "ctypes"])
def test_float32_fused_losses_under_create_graph(dev, golden, host_path):
    """A float32 caller of the FUSED losses (K3 behind the C++ autograd node of the native host path, and behind the
    Python Function of the ctypes path): a plain backward still takes the kernel's gradient; create_graph=True switches
    to the differentiable composition in double.  Compared with the reference evaluated in double on the same
    float32-valued inputs (scenes re-drawn from the seed: the sampler reproduces the reference's draws)."""
    from svbrdf_estimation_amd import _hostext, losses, renderers
    g = golden("g16_second_order.npz")
    _hostext.set_enabled(host_path == "native")
    try:
        if host_path == "native":
            assert _hostext.module() is not None, "native host extension not built"
        tgt = torch.from_numpy(g["f32v_loss_target"]).float().to(dev)
        v = torch.from_numpy(g["loss_v"]).float().to(dev)
        for name, fn in (("loss", losses.RenderingLoss(renderers.LocalRenderer())),
                         ("mixed", losses.MixedLoss(renderers.LocalRenderer()))):
            x = torch.from_numpy(g["f32v_loss_input"]).float().to(dev).requires_grad_(True)
            torch.manual_seed(int(g["loss_rng_seed"]))
            val, grad, pen, hv = _second_order_of(fn, x, tgt, v)
            assert val.dtype == torch.float32 and grad.dtype == torch.float32 and pen.dtype == torch.float32
            ref = float(g["f32v_" + name + "_value"])
            assert abs(val.item() - ref) <= 1e-5 * abs(ref), (name, val.item(), ref)
            _close(grad.cpu().numpy(), g["f32v_" + name + "_grad"], 1e-4, 1e-5, name + ": float32 gradient under create_graph")
            _close(pen.cpu().numpy(), g["f32v_" + name + "_penalty_grad"], 2e-4, 2e-5, name + ": float32 gradient penalty")
            _close(hv.cpu().numpy(), g["f32v_" + name + "_hvp"], 2e-4, 2e-5, name + ": float32 Hessian-vector product")
            # loss.backward(create_graph=True), the other spelling: the accumulated gradient carries a graph
            x3 = x.detach().clone().requires_grad_(True)
            torch.manual_seed(int(g["loss_rng_seed"]))
            fn(x3, tgt).backward(create_graph=True)
            assert x3.grad.requires_grad
            _close(x3.grad.detach().cpu().numpy(), g["f32v_" + name + "_grad"], 1e-4, 1e-5, name + ": backward(create_graph=True)")
            # and the plain backward of the same call is the kernel's gradient, as before
            x2 = x.detach().clone().requires_grad_(True)
            torch.manual_seed(int(g["loss_rng_seed"]))
            fn(x2, tgt).backward()
            _close(x2.grad.cpu().numpy(), g["f32v_" + name + "_grad"], 1e-4, 1e-5, name + ": plain float32 gradient")
    finally:
        _hostext.set_enabled(True)


@pytest.mark.parametrize("host_path", ["native", "ctypes"])
def test_float32_render_and_head_loss_under_create_graph(dev, golden, host_path):
    """``LocalRenderer.render`` on float32 maps (C++ node / Python Function) and the head-fused loss: their
    create_graph=True results against the float64 composition on the same values"""
    from svbrdf_estimation_amd import _hostext, environment as env, losses, renderers
    g = golden("g16_second_order.npz")
    _hostext.set_enabled(host_path == "native")
    try:
        R = renderers.LocalRenderer()
        scene = env.scenes_from_table(torch.from_numpy(g["render_scenes"]))[1]
        maps32 = torch.from_numpy(g["f32v_loss_input"]).float().to(dev)
        w = torch.from_numpy(g["loss_v"]).float().to(dev)

        def hvp_of(m, weight):
            rend = R.render(scene, m)
            (gr,) = torch.autograd.grad((rend * rend).sum(), m, create_graph=True)
            (hv,) = torch.autograd.grad((gr * weight).sum(), m)
            return gr.detach(), hv
        g32, h32 = hvp_of(maps32.clone().requires_grad_(True), w)
        g64, h64 = hvp_of(maps32.double().requires_grad_(True), w.double())
        assert g32.dtype == torch.float32 and h32.dtype == torch.float32
        _close(g32.cpu().numpy(), g64.cpu().numpy(), 1e-4, 1e-5, "float32 render gradient under create_graph")
        _close(h32.cpu().numpy(), h64.cpu().numpy(), 2e-4, 2e-5, "float32 render Hessian-vector product")
        # head-fused loss: [B,9,H,W] encoded input
        gen = torch.Generator().manual_seed(17)
        enc = (torch.rand(2, 9, 12, 12, generator=gen) * 1.6 - 0.8).to(dev)
        tgt = torch.from_numpy(g["f32v_loss_target"]).float().to(dev)
        fn = losses.FusedHeadLoss(R, l1_weight=0.1)
        table = torch.from_numpy(g["f32v_mixed_scenes"])
        x = enc.clone().requires_grad_(True)
        torch.manual_seed(int(g["loss_rng_seed"]))
        val = fn(x, tgt)
        (gr,) = torch.autograd.grad(val, x, create_graph=True)
        (pen,) = torch.autograd.grad((gr ** 2).sum(), x)
        xd = enc.double().requires_grad_(True)
        vald = losses.composed_loss(xd, tgt, table, 0.1, 0.1, 0.01, head=True)
        (grd,) = torch.autograd.grad(vald, xd, create_graph=True)
        (pend,) = torch.autograd.grad((grd ** 2).sum(), xd)
        assert abs(val.item() - vald.item()) <= 1e-5 * abs(vald.item())
        _close(gr.detach().cpu().numpy(), grd.detach().cpu().numpy(), 1e-4, 1e-5, "head loss gradient under create_graph")
        _close(pen.cpu().numpy(), pend.cpu().numpy(), 2e-4, 2e-5, "head loss gradient penalty")
    finally:
        _hostext.set_enabled(True)


@pytest.mark.parametrize("B,H,S,tiled", [(1, 5, 2, True), (3, 8, 4, False), (2, 17, 3, True), (1, 32, 9, False)])
def test_seeded_sweep_second_order_against_the_eager_oracle(dev, B, H, S, tiled):
    """odd sizes, few and many scenes, tied and untied roughness, maps with clamped roughness and back-facing normals: the fused
    float32 loss's create_graph=True gradient penalty against the eager restatement's double backward on the CPU (float64 on
    the same float32 values; the restatement is pinned against the reference's second derivatives in test_oracle_golden.py)"""
    import synth
    from oracle import eager_torch
    from svbrdf_estimation_amd import environment as env, losses, renderers
    seed = 1000 + 7 * H + S
    inp, tgt = synth.make_maps(seed, B, H, tiled_roughness=tiled), synth.make_maps(seed + 1, B, H, tiled_roughness=tiled)
    inp[0, 6:9, 0, :] = 0.0005                                   # below the roughness clamp
    inp[-1, 0:3, -1, :] = np.array([0.8, 0.0, -0.6], np.float32)[:, None]
    torch.manual_seed(seed)
    table = env.BatchSceneSampler(B, 1, S - 1).sample().clone()
    xd = torch.from_numpy(inp.astype(np.float64)).requires_grad_(True)
    val_ref = eager_torch.rendering_loss(xd, torch.from_numpy(tgt.astype(np.float64)), table)
    (g_ref,) = torch.autograd.grad(val_ref, xd, create_graph=True)
    (pen_ref,) = torch.autograd.grad((g_ref ** 2).sum(), xd)
    fn = losses.RenderingLoss(renderers.LocalRenderer())
    fn.random_configuration_count, fn.specular_configuration_count = 1, S - 1
    x = torch.from_numpy(inp).to(dev).requires_grad_(True)
    torch.manual_seed(seed)                                      # the same draws as the table above
    val = fn(x, torch.from_numpy(tgt).to(dev))
    (g,) = torch.autograd.grad(val, x, create_graph=True)
    (pen,) = torch.autograd.grad((g ** 2).sum(), x)
    assert abs(val.item() - val_ref.item()) <= 1e-5 * abs(val_ref.item())
    _close_but(g.detach().cpu().numpy(), g_ref.detach().numpy(), 1e-4, 1e-5, "gradient under create_graph")
    _close_but(pen.cpu().numpy(), pen_ref.numpy(), 2e-4, 2e-5, "gradient penalty")
    assert not pen[0, 6:9, 0].cpu().numpy().any()


def test_an_error_in_the_second_order_hook_reaches_the_caller(dev):
    """the C++ nodes run on autograd's worker thread: an exception raised by the Python code they call back into must come
    out of ``torch.autograd.grad`` as an error, with its message"""
    from svbrdf_estimation_amd import _hostext, losses, renderers
    ext = _hostext.module()
    assert ext is not None, "native host extension not built"

    def broken(*args):
        raise ZeroDivisionError("deliberate failure in the hook")
    ext.set_second_order_hooks(broken, broken)
    try:
        x = (torch.rand(1, 12, 8, 8, device=dev) * 0.5 + 0.25).requires_grad_(True)
        loss = losses.RenderingLoss(renderers.LocalRenderer())(x, x.detach() * 0.9)
        with pytest.raises(RuntimeError, match="deliberate failure in the hook"):
            torch.autograd.grad(loss, x, create_graph=True)
    finally:
        ext.set_second_order_hooks(_hostext._loss_second_order, _hostext._render_second_order)
    loss = losses.RenderingLoss(renderers.LocalRenderer())(x, x.detach() * 0.9)          # and the path still works afterwards
    (g,) = torch.autograd.grad(loss, x, create_graph=True)
    assert g.requires_grad and torch.isfinite(g).all()


def test_third_order_is_refused_loudly_and_the_abi_checks_its_arguments(dev):
    from svbrdf_estimation_amd import _native, renderers
    R = renderers.LocalRenderer()
    table = torch.tensor([[0.2, -0.3, 2.0, 0.5, 0.4, 1.5, 20.0, 30.0, 40.0]], dtype=torch.float32)
    gen = torch.Generator().manual_seed(3)
    m = torch.rand(1, 12, 4, 4, generator=gen, dtype=torch.float64) * 0.5 + 0.25
    m[:, 2] = 1.0
    m = m.to(dev).requires_grad_(True)
    rend = R.render_many(table, m)
    (g1,) = torch.autograd.grad(rend.sum(), m, create_graph=True)
    (g2,) = torch.autograd.grad(g1.sum(), m, create_graph=True)
    assert g2.requires_grad                     # it does depend on the maps ...
    with pytest.raises(RuntimeError, match="third-order derivatives"):
        torch.autograd.grad(g2.sum(), m)        # ... and says so instead of pretending to be a constant
    lib = _native._load()
    p = ctypes.c_void_p
    assert lib.svbrdf_render_bwd_jvp_f64(None, None, None, None, None, None, None, 1, 1, 4, 4, None) == -1
    buf = torch.zeros(1, 12, 4, 4, dtype=torch.float64, device=dev)
    out = torch.zeros(1, 1, 3, 4, 4, dtype=torch.float64, device=dev)
    sc = table.to(dev).view(1, 1, 9).contiguous()
    xr = _native.xrow(dev, 4)
    args = [p(buf.data_ptr()), p(buf.data_ptr()), p(sc.data_ptr()), p(xr.data_ptr()), p(out.data_ptr()), p(buf.data_ptr()),
            p(out.data_ptr())]
    assert lib.svbrdf_render_bwd_jvp_f64(*args, 1, 1, 4, 5, None) == -2                    # H != W
    args[0] = p(buf.data_ptr() + 4)
    assert lib.svbrdf_render_bwd_jvp_f64(*args, 1, 1, 4, 4, None) == -3                    # misaligned double buffer
    with pytest.raises(TypeError):
        _native.render_bwd_jvp_f64(buf.float(), buf, sc, out)
