"""CPU: the C oracle (oracle/svbrdf_oracle.c) against the fixtures generated from the
reference itself (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

import synth
from tolerances import (assert_grad_close, assert_loss_close, assert_render_strict,
                        assert_render_vs_reference)


def _both(oracle, maps, scenes):
    return oracle.render_fwd(maps, scenes), oracle.render_fwd(maps, scenes, f64=True)


@pytest.mark.parametrize("name", ["g1_render_64.npz", "g1_render_32_tiled.npz"])
def test_render_forward_small(oracle, golden, name):
    g = golden(name)
    out, out64 = _both(oracle, g["maps"][None], g["scenes"][None])
    assert_render_vs_reference(out[0], g["out"], out64[0], name)


@pytest.mark.parametrize("H", [256, 512])
def test_render_forward_full_size_lattice(oracle, golden, H):
    g = golden("g2_render_lattice_%d.npz" % H)
    maps = synth.make_maps(int(g["synth_seed"]), 1, H)
    assert synth.checksum(maps[0]) == str(g["maps_sha256"]), "synthetic inputs are not bit-reproducible here"
    st = int(g["stride"])
    out, out64 = _both(oracle, maps, g["scenes"][None])
    assert_render_vs_reference(out[0][:, :, ::st, ::st], g["out_lattice"], out64[0][:, :, ::st, ::st],
                               "lattice %d" % H, scale=float(g["out_max"]))
    sums = out[0].astype(np.float64).sum(axis=(2, 3))
    np.testing.assert_allclose(sums, g["plane_sums"], rtol=2e-6)


def test_kat_bit_exact(oracle, golden):
    """KAT-1 / KAT-2 of SURVEY.md section 4: the oracle reproduces the reference bit for bit."""
    g = golden("g9_kat.npz")
    for k in ("kat1", "kat2"):
        m, sc = g[k + "_maps"][None], g[k + "_scene"][None, None]
        out = oracle.render_fwd(m, sc)[0]
        assert np.array_equal(out, g[k + "_out"]), k
        grad = oracle.render_bwd(m, sc, np.ones((1, 1, 3, 2, 2), np.float32))[0]
        assert_grad_close(grad, g[k + "_grad_of_sum"], k + " grad", rtol=1e-6, afrac=1e-6)
    np.testing.assert_array_equal(g["kat1_out"][0, :, 0, 0].view(np.uint32),
                                  np.array([0x3f86e35a, 0x3f588c83, 0x3f235253], dtype=np.uint32))


@pytest.mark.parametrize("name", ["g3_loss_48.npz", "g3_loss_7_s5.npz", "g3_loss_20_untied.npz"])
def test_rendering_loss_and_gradient(oracle, golden, name):
    g = golden(name)
    loss, grad = oracle.rendering_loss(g["input"], g["target"], g["scenes"])
    assert_loss_close(loss, g["loss"], name)
    assert_grad_close(grad, g["grad_input"], name + " grad_input")
    # forward-only variant returns the same loss
    loss2, none = oracle.rendering_loss(g["input"], g["target"], g["scenes"], want_grad=False)
    assert none is None and loss2 == loss
    # not less accurate than the reference: both measured against the fp64 evaluation
    loss64, grad64 = oracle.rendering_loss(g["input"], g["target"], g["scenes"], f64=True)
    e_or = np.abs(grad - grad64).max()
    e_ref = np.abs(g["grad_input"] - grad64).max()
    assert e_or <= 1.5 * e_ref + 1e-12
    assert abs(loss - loss64) <= 2e-7 * abs(loss64)


@pytest.mark.parametrize("name", ["g3_loss_48.npz", "g3_loss_7_s5.npz", "g3_loss_20_untied.npz"])
def test_mixed_loss_and_gradient(oracle, golden, name):
    """losses.py:54-63 MixedLoss and losses.py:7-19 SVBRDFL1Loss against the reference's values"""
    g = golden(name)
    loss, grad = oracle.mixed_loss(g["input"], g["target"], g["scenes"], 0.1)
    assert_loss_close(loss, g["mixed_loss"], name)
    assert_grad_close(grad, g["mixed_grad"], name + " mixed grad")
    l1, gl1 = oracle.mixed_loss(g["input"], g["target"], g["scenes"], 1.0)
    r, gr = oracle.rendering_loss(g["input"], g["target"], g["scenes"])
    assert_loss_close(l1 - r, g["l1_loss"], name + " l1")
    assert_grad_close(gl1 - gr, g["l1_grad"], name + " l1 grad")


def test_edge_cases(oracle, golden):
    g = golden("g4_edge_cases.npz")
    for name in g["names"]:
        m, sc = g[name + "__maps"][None], g[name + "__scene"][None, None]
        out, out64 = _both(oracle, m, sc)
        if np.abs(g[name + "__out"]).max() == 0:
            assert np.array_equal(out[0], g[name + "__out"]), name
        else:
            assert_render_vs_reference(out[0], g[name + "__out"], out64[0], name)
        grad = oracle.render_bwd(m, sc, g[name + "__cot"][None])[0]
        if np.abs(g[name + "__grad"]).max() == 0:
            assert not grad.any(), name
        else:
            assert_grad_close(grad, g[name + "__grad"], name + " grad")
    # clamp masks: roughness below 1e-3 gets exactly zero gradient, AT the clamp it passes
    m, sc = g["r_below_clamp__maps"][None], g["r_below_clamp__scene"][None, None]
    grad = oracle.render_bwd(m, sc, g["r_below_clamp__cot"][None])[0]
    H = m.shape[-1]
    assert not grad[7:9, :, : H // 2].any()
    assert grad[6, 0, 0] != 0 and grad[6, 0, 0] == pytest.approx(g["r_below_clamp__grad"][6, 0, 0], rel=1e-4)
    assert not grad[6, 1:, : H // 2].any()


def test_batched_one_scene_and_odd_sizes(oracle, golden):
    g = golden("g4_batched_one_scene.npz")
    sc = np.repeat(g["scene"][None, None], g["maps"].shape[0], 0)
    out, out64 = _both(oracle, g["maps"], sc)
    assert_render_vs_reference(out[:, 0], g["out"], out64[:, 0], "batched")
    assert_grad_close(oracle.render_bwd(g["maps"], sc, g["cot"][:, None]), g["grad"], "batched grad")
    assert tuple(g["out3_shape"]) == (1, 3, 16, 16)
    g = golden("g4_render_7x7.npz")
    sc = np.repeat(g["scenes"][None], 2, 0)
    out, out64 = _both(oracle, g["maps"], sc)
    assert_render_vs_reference(out, g["out"], out64, "7x7")


def test_linspace_bits(oracle, golden):
    g = golden("g6_linspace.npz")
    for k in g.files:
        W = int(k[2:])
        assert np.array_equal(oracle.make_xrow(W).view(np.uint32), g[k].view(np.uint32)), W


def test_render_bwd_is_adjoint_of_fwd(oracle):
    """<J v, w> == <v, J^T w> by central differences in fp64 (independent of the reference)."""
    maps = synth.make_maps(9, 1, 6, tiled_roughness=False, r_lo=0.2)
    sc = np.array([[[0.3, -0.4, 1.5, -0.5, 0.6, 1.2, 20.0, 25.0, 30.0]]], np.float32)
    w = (synth.uniform01(77, (1, 1, 3, 6, 6)) - 0.5).astype(np.float32)
    gt = oracle.render_bwd(maps, sc, w, f64=True)
    v = (synth.uniform01(78, maps.shape) - 0.5).astype(np.float32)
    h = np.float32(2.0 ** -10)
    op = oracle.render_fwd(maps + h * v, sc, f64=True)
    om = oracle.render_fwd(maps - h * v, sc, f64=True)
    lhs = ((op - om) / (2.0 * float(h)) * w).sum()
    rhs = (gt * v).sum()
    assert lhs == pytest.approx(rhs, rel=2e-4)


def test_bad_dims_rejected(oracle):
    with pytest.raises(RuntimeError):
        oracle.render_fwd(np.zeros((1, 12, 4, 8), np.float32), np.zeros((1, 1, 9), np.float32),
                          xrow=np.zeros(8, np.float32))


def test_head_decode_and_head_loss(oracle, golden):
    """row f1: oracle restatement of the model head (models.py:338-346) + mixed loss vs the reference"""
    g = golden("g11_head_loss.npz")
    np.testing.assert_allclose(oracle.head_decode(g["enc9"]), g["decoded12"], rtol=3e-7, atol=1e-7)
    for tag, w in (("mixed", 0.1), ("render", 0.0)):
        loss, grad = oracle.head_loss(g["enc9"], g["target"], g["scenes"], w)
        _, g64 = oracle.head_loss(g["enc9"], g["target"], g["scenes"], w, f64=True)
        assert_loss_close(loss, g[tag + "_loss"], tag)
        assert_grad_close(grad, g[tag + "_grad9"], tag + " grad9", f64=g64)


def test_eager_restatement_against_the_reference_first_and_second_order(golden):
    """oracle/eager_torch.py -- the reference's algorithm as eager PyTorch, which bench.py times as the CPU baseline -- on
    the reference's own outputs: the float32 loss and gradient of g3, and in float64 (the reference's mixed precision for
    double maps) the second-order fixture g16: gradient under create_graph, gradient of the penalty sum(g^2), Hessian-
    vector product.  Same ops on the same machine class: the float64 results agree to rounding."""
    import torch
    from oracle import eager_torch
    g = golden("g3_loss_48.npz")
    x = torch.from_numpy(g["input"]).clone().requires_grad_(True)
    loss = eager_torch.rendering_loss(x, torch.from_numpy(g["target"]), torch.from_numpy(g["scenes"]))
    loss.backward()
    assert_loss_close(loss.item(), g["loss"], "eager loss")
    assert_grad_close(x.grad.numpy(), g["grad_input"], "eager grad_input")
    s = golden("g16_second_order.npz")
    x = torch.from_numpy(s["loss_input"]).clone().requires_grad_(True)
    val = eager_torch.rendering_loss(x, torch.from_numpy(s["loss_target"]), torch.from_numpy(s["loss_scenes"]))
    assert val.dtype == torch.float64 and abs(val.item() - float(s["loss_value"])) <= 1e-12 * abs(float(s["loss_value"]))
    (gr,) = torch.autograd.grad(val, x, create_graph=True)
    (pen,) = torch.autograd.grad((gr ** 2).sum(), x, retain_graph=True)
    (hv,) = torch.autograd.grad((gr * torch.from_numpy(s["loss_v"])).sum(), x)
    for got, key in ((gr.detach(), "loss_grad"), (pen, "loss_penalty_grad"), (hv, "loss_hvp")):
        ref = s[key]
        assert np.abs(got.numpy() - ref).max() <= 1e-9 * np.abs(ref).max(), key
