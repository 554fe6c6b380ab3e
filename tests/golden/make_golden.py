#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, never on the GPU box):

    python tests/golden/make_golden.py

The reference (mworchel/svbrdf-estimation, pure Python/PyTorch) is imported
read-only from /root/reference/development/multiImage_pytorch with byte-code
writing disabled.  `cv2` and `pyredner` are absent from this image and are used
only by OrthoToPerspectiveMapping / RednerRenderer (off the hot path), so empty
placeholder modules are registered for the two top-level imports
(renderers.py:1,4); nothing from them is ever called.

The fixtures are DATA ONLY: inputs (or the seed of tests/synth.py that
regenerates them + a sha256 of the regenerated array) and the reference's
outputs.  Environment of the run is recorded in MANIFEST.json.

NOTE on reproducibility of the reference: torch's CPU `sqrt`/`log` go through
MKL VML for larger tensors and are NOT correctly rounded (0.7 % of results are
1 ULP off, measured), and the GGX denominator (renderers.py:26) amplifies a
1-ULP change of NH by up to 1e3..1e4.  The reference is therefore only
reproducible across its own backends to ~1e-4 relative at highlight pixels;
the tolerances in tests/  account for that.
"""
import json
import math
import os
import platform
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

REF = "/root/reference/development/multiImage_pytorch"
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.modules.setdefault("pyredner", types.ModuleType("pyredner"))
sys.path.insert(0, REF)

import torch  # noqa: E402

import environment as ref_env  # noqa: E402
import losses as ref_losses  # noqa: E402
import renderers as ref_renderers  # noqa: E402
import utils as ref_utils  # noqa: E402


def scene_row(sc):
    f = lambda v: np.asarray(torch.as_tensor(v, dtype=torch.float32).numpy(), dtype=np.float32)
    return np.concatenate([f(sc.camera.pos), f(sc.light.pos), f(sc.light.color)])


def scene_table(scenes):
    return np.stack([scene_row(s) for s in scenes]).astype(np.float32)


def sample_scenes(seed, n_random=3, n_specular=6):
    torch.manual_seed(seed)
    return ref_env.generate_random_scenes(n_random) + ref_env.generate_specular_scenes(n_specular)


def render_all(maps_t, scenes):
    R = ref_renderers.LocalRenderer()
    return torch.cat([R.render(sc, maps_t) for sc in scenes], dim=0)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KiB" % (name, os.path.getsize(path) / 1024.0))


def g1_render_64():
    maps = synth.make_maps(101, 1, 64, tiled_roughness=False)[0]
    scenes = sample_scenes(7)
    out = render_all(torch.from_numpy(maps), scenes).numpy()
    save("g1_render_64.npz", maps=maps, scenes=scene_table(scenes), out=out)
    # model-like variant: single roughness channel tiled x3, smaller
    maps = synth.make_maps(102, 1, 32, tiled_roughness=True)[0]
    out = render_all(torch.from_numpy(maps), scenes).numpy()
    save("g1_render_32_tiled.npz", maps=maps, scenes=scene_table(scenes), out=out)


def g2_lattice(H, stride, seed):
    maps = synth.make_maps(seed, 1, H, tiled_roughness=True)[0]
    scenes = sample_scenes(313)
    out = render_all(torch.from_numpy(maps), scenes).numpy()
    save("g2_render_lattice_%d.npz" % H,
         synth_seed=np.int64(seed), H=np.int64(H), stride=np.int64(stride),
         maps_sha256=np.array(synth.checksum(maps)),
         scenes=scene_table(scenes),
         out_lattice=out[:, :, ::stride, ::stride].copy(),
         out_max=np.float32(out.max()),
         plane_sums=out.astype(np.float64).sum(axis=(2, 3)))


class _Recorder:
    """records the scenes RenderingLoss.forward draws (losses.py:35)"""

    def __init__(self):
        self.items = []
        self._r, self._s = ref_env.generate_random_scenes, ref_env.generate_specular_scenes

    def __enter__(self):
        def rnd(n):
            sc = self._r(n)
            self.items.append(sc)
            return sc

        def spc(n):
            sc = self._s(n)
            self.items[-1] = self.items[-1] + sc
            return sc
        ref_env.generate_random_scenes, ref_env.generate_specular_scenes = rnd, spc
        return self

    def __exit__(self, *a):
        ref_env.generate_random_scenes, ref_env.generate_specular_scenes = self._r, self._s

    def table(self):
        return np.stack([scene_table(s) for s in self.items])


def g3_loss(name, B, H, seed, rng_seed, n_random=3, n_specular=6, tiled=True):
    inp = synth.make_maps(seed, B, H, tiled_roughness=tiled)
    tgt = synth.make_maps(seed + 1, B, H, tiled_roughness=tiled)
    x = torch.from_numpy(inp).clone().requires_grad_(True)
    loss_fn = ref_losses.RenderingLoss(ref_renderers.LocalRenderer())
    loss_fn.random_configuration_count = n_random
    loss_fn.specular_configuration_count = n_specular
    torch.manual_seed(rng_seed)
    with _Recorder() as rec:
        loss = loss_fn(x, torch.from_numpy(tgt))
    loss.backward()
    # G7: SVBRDFL1Loss / MixedLoss on the same inputs (same RNG seed for the mixed loss)
    x2 = torch.from_numpy(inp).clone().requires_grad_(True)
    l1 = ref_losses.SVBRDFL1Loss()(x2, torch.from_numpy(tgt))
    l1.backward()
    x3 = torch.from_numpy(inp).clone().requires_grad_(True)
    torch.manual_seed(rng_seed)
    mixed_fn = ref_losses.MixedLoss(ref_renderers.LocalRenderer())
    mixed_fn.rendering_loss.random_configuration_count = n_random
    mixed_fn.rendering_loss.specular_configuration_count = n_specular
    mixed = mixed_fn(x3, torch.from_numpy(tgt))
    mixed.backward()
    save(name, input=inp, target=tgt, scenes=rec.table(), rng_seed=np.int64(rng_seed),
         n_random=np.int64(n_random), n_specular=np.int64(n_specular),
         loss=np.float32(loss.item()), grad_input=x.grad.numpy(),
         l1_loss=np.float32(l1.item()), l1_grad=x2.grad.numpy(),
         mixed_loss=np.float32(mixed.item()), mixed_grad=x3.grad.numpy())


def g4_edge_cases():
    """small crafted patches; each case: maps [B,12,H,W], one scene, out, cotangent, grad"""
    H = 16
    cases = {}

    def base(seed, **kw):
        return synth.make_maps(seed, 1, H, **kw)[0]

    cam_top, light_top = [0.0, 0.0, 2.0], [0.3, -0.2, 1.5]
    # (a) roughness below the 1e-3 clamp on half the patch: zero roughness gradient there
    m = base(201, tiled_roughness=False)
    m[6:9, :, : H // 2] = np.float32(0.0005)
    m[6, 0, 0] = np.float32(0.001)  # exactly AT the clamp: inclusive mask passes the gradient
    cases["r_below_clamp"] = (m, cam_top, light_top, [20.0, 20.0, 20.0])
    # (b) normals facing away from the light: radiance 0, LN clamp path, zero gradient via LN+
    m = base(202)
    m[0] = np.float32(0.9)
    m[1] = np.float32(0.0)
    m[2] = np.float32(0.1)
    cases["n_dot_wi_negative"] = (m, [0.2, 0.1, 1.0], [-3.0, 0.0, 0.2], [50.0, 50.0, 50.0])
    # (c) GGX denominator clamp active: tiny roughness, mirror configuration
    m = base(203)
    m[0:2] = np.float32(0.0)
    m[2] = np.float32(1.0)
    m[6:9] = np.float32(0.01)
    cases["den_clamp_active"] = (m, [0.5, 0.5, 2.0], [-0.5, -0.5, 2.0], [50.0, 50.0, 50.0])
    # (d) grazing camera (z ~ 0.05): VN clamp
    cases["grazing_camera"] = (base(204, tiled_roughness=False), [1.8, 0.3, 0.05], [0.1, 0.4, 1.2], [20.0, 20.0, 20.0])
    # (e) non-unit normals (the renderer never normalises n)
    m = base(205, unit_normals=False, tiled_roughness=False)
    m[0:3] *= np.float32(1.7)
    cases["non_unit_normals"] = (m, [-0.4, 0.6, 1.1], [0.7, 0.2, 0.9], [20.0, 30.0, 40.0])
    # (f) light inside the patch plane neighbourhood, coloured light
    cases["near_light"] = (base(206), [0.0, 0.0, 1.0], [0.2, -0.3, 0.05], [5.0, 1.0, 0.2])

    R = ref_renderers.LocalRenderer()
    arrays = {"names": np.array(sorted(cases))}
    for idx, name in enumerate(sorted(cases)):
        m, cam, light, col = cases[name]
        sc = ref_env.Scene(ref_env.Camera(cam), ref_env.Light(light, col))
        x = torch.from_numpy(m).clone().requires_grad_(True)
        out = R.render(sc, x)
        cot = synth.uniform01(300 + idx, tuple(out.shape)) - np.float32(0.5)
        out.backward(torch.from_numpy(cot))
        arrays[name + "__maps"] = m
        arrays[name + "__scene"] = scene_row(sc)
        arrays[name + "__out"] = out.detach().numpy()
        arrays[name + "__cot"] = cot
        arrays[name + "__grad"] = x.grad.numpy()
    save("g4_edge_cases.npz", **arrays)

    # (g) 4-D batched input with ONE scene for the whole batch, tensor-valued positions and
    #     tensor-valued light colour (the dataloader's call shape, dataset.py:206-212)
    mb = synth.make_maps(207, 3, H, tiled_roughness=False)
    sc = ref_env.Scene(ref_env.Camera(torch.tensor([0.1, -0.2, 2.75])),
                       ref_env.Light(torch.tensor([0.4, 0.3, 2.197]), torch.tensor([28.0, 30.0, 31.5])))
    x = torch.from_numpy(mb).clone().requires_grad_(True)
    out = R.render(sc, x)
    cot = synth.uniform01(777, tuple(out.shape)) - np.float32(0.5)
    out.backward(torch.from_numpy(cot))
    # 3-D input -> [1,3,H,W] output shape quirk (renderers.py:98)
    out3 = R.render(sc, torch.from_numpy(mb[0]))
    save("g4_batched_one_scene.npz", maps=mb, scene=scene_row(sc), out=out.detach().numpy(),
         cot=cot, grad=x.grad.numpy(), out3_shape=np.array(out3.shape), out3=out3.numpy())
    # odd, tiny, non-multiple-of-anything size
    m7 = synth.make_maps(208, 2, 7, tiled_roughness=False)
    scenes = sample_scenes(5)
    outs = np.stack([render_all(torch.from_numpy(m7[b]), scenes).numpy() for b in range(2)])
    save("g4_render_7x7.npz", maps=m7, scenes=scene_table(scenes), out=outs)


def g5_sampler():
    arrays = {}
    for seed in (0, 7, 313):
        arrays["seed_%d" % seed] = scene_table(sample_scenes(seed))
    # two consecutive items from one RNG stream (what RenderingLoss does for B=2)
    torch.manual_seed(99)
    a = ref_env.generate_random_scenes(3) + ref_env.generate_specular_scenes(6)
    b = ref_env.generate_random_scenes(3) + ref_env.generate_specular_scenes(6)
    arrays["seed_99_two_items"] = np.stack([scene_table(a), scene_table(b)])
    arrays["seed_5_11_21"] = scene_table(sample_scenes(5, 11, 21))
    torch.manual_seed(3)
    arrays["seed_3_dirs_8"] = ref_utils.generate_normalized_random_direction(8, 0.001, 0.05).numpy()
    save("g5_scene_sampler.npz", **arrays)


def g6_linspace():
    save("g6_linspace.npz", **{"W_%d" % W: torch.linspace(-1, 1, W).numpy() for W in (2, 3, 7, 48, 64, 100, 256, 512)})


def g8_utils():
    x = synth.uniform01(401, (2, 9, 32, 32)) * np.float32(2.0) - np.float32(1.0)
    t = torch.from_numpy(x).clone().requires_grad_(True)
    dec = ref_utils.decode_svbrdf(t)
    cot = synth.uniform01(402, tuple(dec.shape)) - np.float32(0.5)
    dec.backward(torch.from_numpy(cot))
    y = synth.uniform01(403, (2, 3, 8, 8)) * np.float32(2.0)
    n, d, r, s = ref_utils.unpack_svbrdf(torch.from_numpy(synth.make_maps(404, 2, 8)))
    save("g8_utils.npz", enc9=x, decoded12=dec.detach().numpy(), cot=cot, grad9=t.grad.numpy(),
         decoded12_single=ref_utils.decode_svbrdf(torch.from_numpy(x[0])).numpy(),
         img=y, gamma_enc=ref_utils.gamma_encode(torch.from_numpy(y)).numpy(),
         gamma_dec=ref_utils.gamma_decode(torch.from_numpy(y)).numpy(),
         unit=ref_utils.encode_as_unit_interval(torch.from_numpy(x)).numpy(),
         from_unit=ref_utils.decode_from_unit_interval(torch.from_numpy(y)).numpy(),
         maps404=synth.make_maps(404, 2, 8), n=n.numpy(), d=d.numpy(), r=r.numpy(), s=s.numpy(),
         repacked=ref_utils.pack_svbrdf(n, d, r, s).numpy(),
         # the reference's own unit-test constants, utils.py:153-157 and :172-178
         magic_pixel=np.float64(1.3703509847201), magic_decoded=np.float64(2.0))


def g10_render_inputs():
    """dataset.py:162-221 render_inputs: scenes (captured at the renderer call) and final photos"""
    import dataset as ref_dataset

    class FakeSelf:          # render_inputs only reads self.use_augmentation
        pass
    captured = []
    orig = ref_renderers.LocalRenderer.render

    def spy(self_r, scene, svbrdf):
        captured.append(scene_row(scene))
        return orig(self_r, scene, svbrdf)
    arrays = {}
    ref_renderers.LocalRenderer.render = spy
    try:
        for aug in (False, True):
            for count in (1, 4):
                maps = synth.make_maps(500 + count, 1, 32)[0]
                fs = FakeSelf()
                fs.use_augmentation = aug
                del captured[:]
                torch.manual_seed(40 + count + (100 if aug else 0))
                out = ref_dataset.SvbrdfDataset.render_inputs(fs, torch.from_numpy(maps), count)
                key = "aug%d_n%d" % (int(aug), count)
                arrays[key + "__maps"] = maps
                arrays[key + "__scenes"] = np.stack(captured)
                arrays[key + "__out"] = out.numpy()
                arrays[key + "__seed"] = np.int64(40 + count + (100 if aug else 0))
                arrays[key + "__rng_after"] = torch.get_rng_state().numpy()[:64].copy()
    finally:
        ref_renderers.LocalRenderer.render = orig
    save("g10_render_inputs.npz", **arrays)


def g11_head_loss():
    """row f1: the model head (models.py:338-346) + MixedLoss, gradient w.r.t. the 9 encoded channels"""
    B, H = 2, 24
    enc = (synth.uniform01(601, (B, 9, H, H)) * np.float32(1.8) - np.float32(0.9)).astype(np.float32)
    enc[:, 0:2] *= np.float32(0.25)                          # moderate normal tilt, like a trained net
    tgt = synth.make_maps(602, B, H, tiled_roughness=True)

    def head(t):
        sv = ref_utils.decode_svbrdf(t)
        n, d, r, s = ref_utils.unpack_svbrdf(sv)
        return ref_utils.pack_svbrdf(n, ref_utils.encode_as_unit_interval(d), ref_utils.encode_as_unit_interval(r),
                                     ref_utils.encode_as_unit_interval(s))
    out = {}
    for tag, w in (("mixed", 0.1), ("render", 0.0)):
        x = torch.from_numpy(enc).clone().requires_grad_(True)
        maps = head(x)
        torch.manual_seed(17)
        with _Recorder() as rec:
            if w:
                loss = ref_losses.MixedLoss(ref_renderers.LocalRenderer(), l1_weight=w)(maps, torch.from_numpy(tgt))
            else:
                loss = ref_losses.RenderingLoss(ref_renderers.LocalRenderer())(maps, torch.from_numpy(tgt))
        loss.backward()
        out[tag + "_loss"] = np.float32(loss.item())
        out[tag + "_grad9"] = x.grad.numpy()
        out["scenes"] = rec.table()
        out["decoded12"] = maps.detach().numpy()
    save("g11_head_loss.npz", enc9=enc, target=tgt, rng_seed=np.int64(17), **out)


def g9_kat():
    R = ref_renderers.LocalRenderer()
    out = {}
    m = np.zeros((12, 2, 2), np.float32)
    m[2] = 1
    m[3], m[4], m[5] = 0.5, 0.4, 0.3
    m[6:9] = 0.5
    m[9:12] = 0.04
    sc = ref_env.Scene(ref_env.Camera([0.0, 0.0, 2.0]), ref_env.Light([0.0, 0.0, 2.0], [50.0, 50.0, 50.0]))
    x = torch.from_numpy(m).clone().requires_grad_(True)
    o = R.render(sc, x)
    o.sum().backward()
    out.update(kat1_maps=m, kat1_scene=scene_row(sc), kat1_out=o.detach().numpy(), kat1_grad_of_sum=x.grad.numpy())
    m2 = np.zeros((12, 2, 2), np.float32)
    m2[0], m2[1], m2[2] = 0.2, -0.1, 0.97
    m2[3], m2[4], m2[5] = 0.5, 0.4, 0.3
    m2[6], m2[7], m2[8] = 0.3, 0.35, 0.4
    m2[9], m2[10], m2[11] = 0.04, 0.5, 0.9
    sc2 = ref_env.Scene(ref_env.Camera([0.3, -1.0, 2.0]), ref_env.Light([0.5, 0.2, 1.5], [20.0, 30.0, 40.0]))
    x = torch.from_numpy(m2).clone().requires_grad_(True)
    o = R.render(sc2, x)
    o.sum().backward()
    out.update(kat2_maps=m2, kat2_scene=scene_row(sc2), kat2_out=o.detach().numpy(), kat2_grad_of_sum=x.grad.numpy())
    save("g9_kat.npz", **out)


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    if len(sys.argv) > 1 and sys.argv[1] == "--only-untied-loss":   # added after the first freeze
        g3_loss("g3_loss_20_untied.npz", 2, 20, 141, 13, tiled=False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-render-inputs":  # row f3, added later
        g10_render_inputs()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-head-loss":      # row f1, added later
        g11_head_loss()
        return
    g1_render_64()
    g2_lattice(256, 8, 111)
    g2_lattice(512, 16, 112)
    g3_loss("g3_loss_48.npz", 2, 48, 121, 11)
    g3_loss("g3_loss_7_s5.npz", 3, 7, 131, 12, n_random=2, n_specular=3)
    g3_loss("g3_loss_20_untied.npz", 2, 20, 141, 13, tiled=False)   # three independent roughness channels
    g4_edge_cases()
    g5_sampler()
    g6_linspace()
    g8_utils()
    g9_kat()
    g10_render_inputs()
    g11_head_loss()
    manifest = {
        "generator": "tests/golden/make_golden.py",
        "reference": "mworchel/svbrdf-estimation @ /root/reference (development/multiImage_pytorch)",
        "torch": torch.__version__, "numpy": np.__version__, "python": platform.python_version(),
        "cpu_capability": torch.backends.cpu.get_cpu_capability(),
        "machine": platform.machine(),
        "pi_f32": float(np.float32(math.pi)),
    }
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest))


if __name__ == "__main__":
    main()
