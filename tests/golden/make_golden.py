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



def _write_png(path, rows_hw3_uint8):
    from PIL import Image
    Image.fromarray(rows_hw3_uint8, mode="RGB").save(path)


def g12_dataset_reader():
    """row f4: the reference's SvbrdfDataset.read_sample / __getitem__ (dataset.py:44-140: chunk on width, normals
    *2-1, the LAST n photos, scale modes 'crop' (anchor 0 or random) and 'resize' (centre crop + bilinear), gamma
    decode) and mix (dataset.py:142-160), run on small tiled PNGs that are committed next to the outputs.
      g12_tiled_synthetic.png   3 photos + 4 maps, tiles 56 wide x 40 high (landscape), random 8-bit content
      g12_tiled_toy_crop.png    the top-left 64x64 of each of the 14 tiles of the reference's bundled toy sample
                                data/train/10_1_parquet_floor_0.png (10 photos + 4 maps): real material statistics
      g12_maps_only_{0,1}.png   4 map tiles of 32x32 (input_image_count = 0: the material-mixing configuration)"""
    import random
    import shutil
    import tempfile
    from PIL import Image
    import dataset as ref_dataset

    rng = np.random.RandomState(1234)
    syn = rng.randint(0, 256, size=(40, 56 * 7, 3)).astype(np.uint8)
    _write_png(os.path.join(HERE, "g12_tiled_synthetic.png"), syn)
    toy = np.asarray(Image.open(os.path.join(REF, "data", "train", "10_1_parquet_floor_0.png")).convert("RGB"))
    tw = toy.shape[1] // 14
    crop = np.concatenate([toy[:64, k * tw:k * tw + 64] for k in range(14)], axis=1)
    _write_png(os.path.join(HERE, "g12_tiled_toy_crop.png"), np.ascontiguousarray(crop))
    for k in range(2):
        m = rng.randint(0, 256, size=(32, 32 * 4, 3)).astype(np.uint8)
        m[:, :32, 2] = np.maximum(m[:, :32, 2], 160)          # normals: mostly upward z ...
        m[:4, :32, 2] = 128                                   # ... with a band at z ~ 0.004 (< the 0.01 floor of mix)
        m[:, 64:96, 1] = m[:, 64:96, 0]
        m[:, 64:96, 2] = m[:, 64:96, 0]                       # grey roughness, as the data stores it
        _write_png(os.path.join(HERE, "g12_maps_only_%d.png" % k), m)

    arrays = {}

    def dataset_for(png, **kw):
        d = tempfile.mkdtemp()
        shutil.copy(os.path.join(HERE, png), d)
        return ref_dataset.SvbrdfDataset(data_directory=d, use_augmentation=False, **kw), d

    for tag, png, n in (("syn", "g12_tiled_synthetic.png", 3), ("toy", "g12_tiled_toy_crop.png", 10)):
        size = 32
        ds, d = dataset_for(png, image_size=size, scale_mode="crop", input_image_count=n, used_input_image_count=2)
        photos, svbrdf = ds.read_sample(ds.file_paths[0])
        arrays[tag + "__read_photos"] = photos.numpy()
        arrays[tag + "__read_svbrdf"] = svbrdf.numpy()
        item = ds[0]
        arrays[tag + "__crop0_inputs"], arrays[tag + "__crop0_svbrdf"] = item["inputs"].numpy(), item["svbrdf"].numpy()
        ds.random_crop = True
        np.random.seed(5)
        item = ds[0]
        arrays[tag + "__randcrop_inputs"], arrays[tag + "__randcrop_svbrdf"] = item["inputs"].numpy(), item["svbrdf"].numpy()
        arrays[tag + "__randcrop_np_seed"] = np.int64(5)
        ds.random_crop = False
        ds.is_linear = True
        arrays[tag + "__crop0_linear_inputs"] = ds[0]["inputs"].numpy()
        ds.is_linear = False
        ds.scale_mode, ds.image_size = "resize", 24
        item = ds[0]
        arrays[tag + "__resize_inputs"], arrays[tag + "__resize_svbrdf"] = item["inputs"].numpy(), item["svbrdf"].numpy()
        ds.used_input_image_count = 1
        ds.scale_mode, ds.image_size = "crop", size
        arrays[tag + "__crop0_used1_inputs"] = ds[0]["inputs"].numpy()
        shutil.rmtree(d)

    # mix (dataset.py:142-160): explicit alpha, and alpha drawn from the torch generator
    fs = types.SimpleNamespace()
    a = synth.make_maps(701, 1, 32)[0]
    b = synth.make_maps(702, 1, 32, tilt=0.6)[0]
    a[2, :3, :] = np.float32(0.004)                           # below the 0.01 floor of the projection
    b[2, 5, :] = np.float32(-0.2)
    arrays["mix__a"], arrays["mix__b"] = a, b
    arrays["mix__alpha03"] = ref_dataset.SvbrdfDataset.mix(fs, torch.from_numpy(a), torch.from_numpy(b),
                                                           alpha=torch.tensor([0.3])).numpy()
    torch.manual_seed(9)
    arrays["mix__drawn"] = ref_dataset.SvbrdfDataset.mix(fs, torch.from_numpy(a), torch.from_numpy(b)).numpy()
    torch.manual_seed(9)
    arrays["mix__drawn_alpha"] = torch.Tensor(1).uniform_(0.1, 0.9).numpy()
    arrays["mix__seed"] = np.int64(9)

    # __getitem__ with material mixing (input_image_count = 0; no photos requested, so nothing is rendered):
    # python's random picks the partner, torch's generator the blend weight
    d = tempfile.mkdtemp()
    for k in range(2):
        shutil.copy(os.path.join(HERE, "g12_maps_only_%d.png" % k), d)
    ds = ref_dataset.SvbrdfDataset(data_directory=d, image_size=24, scale_mode="crop", input_image_count=0,
                                   used_input_image_count=0, use_augmentation=False, mix_materials=True)
    ds.file_paths = sorted(ds.file_paths)                     # os.listdir order is arbitrary: pin it
    picked = []
    orig = random.randrange

    def spy(lo, hi):
        v = orig(lo, hi)
        picked.append(v)
        return v
    random.randrange = spy
    try:
        for idx in (0, 1):
            random.seed(3 + idx)
            torch.manual_seed(21 + idx)
            item = ds[idx]
            arrays["mixitem%d__svbrdf" % idx] = item["svbrdf"].numpy()
            arrays["mixitem%d__inputs_shape" % idx] = np.array(item["inputs"].shape, np.int64)
            arrays["mixitem%d__partner" % idx] = np.int64(picked[-1])
            arrays["mixitem%d__rng_after" % idx] = torch.get_rng_state().numpy()[:64].copy()
    finally:
        random.randrange = orig
        shutil.rmtree(d)

    # round 3: __getitem__ with material mixing AND scale_mode='resize' (dataset.py:52-73): the reference blends the two
    # materials at full resolution, THEN centre-crops (landscape tiles: 40 wide x 28 high) and resizes bilinearly --
    # the renormalisation of the blended normals does not commute with the resize, so the order is part of the contract
    rng2 = np.random.RandomState(4321)
    for k in range(2):
        m = rng2.randint(0, 256, size=(28, 40 * 4, 3)).astype(np.uint8)
        m[:, :40, 2] = np.maximum(m[:, :40, 2], 150)
        m[:, 80:120, 1] = m[:, 80:120, 0]
        m[:, 80:120, 2] = m[:, 80:120, 0]
        _write_png(os.path.join(HERE, "g12_maps_wide_%d.png" % k), m)
    d = tempfile.mkdtemp()
    for k in range(2):
        shutil.copy(os.path.join(HERE, "g12_maps_wide_%d.png" % k), d)
    ds = ref_dataset.SvbrdfDataset(data_directory=d, image_size=20, scale_mode="resize", input_image_count=0,
                                   used_input_image_count=0, use_augmentation=False, mix_materials=True)
    ds.file_paths = sorted(ds.file_paths)
    picked = []
    random.randrange = spy
    try:
        for idx in (0, 1):
            random.seed(13 + idx)
            torch.manual_seed(31 + idx)
            item = ds[idx]
            arrays["mixresize%d__svbrdf" % idx] = item["svbrdf"].numpy()
            arrays["mixresize%d__inputs_shape" % idx] = np.array(item["inputs"].shape, np.int64)
            arrays["mixresize%d__partner" % idx] = np.int64(picked[-1])
            arrays["mixresize%d__rng_after" % idx] = torch.get_rng_state().numpy()[:64].copy()
    finally:
        random.randrange = orig
        shutil.rmtree(d)
    save("g12_dataset_reader.npz", **arrays)


def _deterministic_state(ref_model, seed0):
    """state dict with the reference's keys/shapes, values from tests/synth.py (platform-independent): each tensor
    uniform with the mean and standard deviation the reference's own initialisation gave it"""
    state, stats = {}, []
    for i, (k, v) in enumerate(ref_model.state_dict().items()):
        mean = np.float32(v.float().mean().item())
        std = np.float32(v.float().std().item()) if v.numel() > 1 else np.float32(0.0)
        u = synth.uniform01(seed0 + i, tuple(v.shape)) - np.float32(0.5)
        w = (u * (np.float32(3.4641016) * std) + mean).astype(np.float32)
        state[k] = torch.from_numpy(w.reshape(tuple(v.shape)))
        stats.append((k, tuple(v.shape), float(mean), float(std)))
    return state, stats


def g13_unet_forward():
    """row f4: forward pass of the reference's SingleViewModel / MultiViewModel (models.py:322-411) with a state dict
    the test can regenerate bit for bit (tests/synth.py values, keyed by the reference's parameter names), outputs
    on a stride-8 lattice.  Lets the GPU box check the MIOpen forward of the re-stated network."""
    import models as ref_models
    arrays = {}
    for tag, cls, shape, seed0 in (("single", ref_models.SingleViewModel, (1, 3, 256, 256), 9000),
                                   ("multi", ref_models.MultiViewModel, (1, 2, 3, 256, 256), 9500)):
        torch.manual_seed(0)
        model = cls(use_coords=True).eval()
        state, stats = _deterministic_state(model, seed0)
        model.load_state_dict(state)
        x = synth.uniform01(seed0 + 400, shape)
        with torch.no_grad():
            y = model(torch.from_numpy(x)).numpy()
        arrays[tag + "__keys"] = np.array([k for k, _, _, _ in stats])
        arrays[tag + "__shapes"] = np.array([",".join(map(str, sh)) for _, sh, _, _ in stats])
        arrays[tag + "__mean"] = np.array([m for _, _, m, _ in stats], np.float32)
        arrays[tag + "__std"] = np.array([sd for _, _, _, sd in stats], np.float32)
        arrays[tag + "__seed0"] = np.int64(seed0)
        arrays[tag + "__input_seed"] = np.int64(seed0 + 400)
        arrays[tag + "__input_shape"] = np.array(shape, np.int64)
        arrays[tag + "__out_lattice"] = y[:, :, ::8, ::8].copy()
        arrays[tag + "__out_plane_sums"] = y.astype(np.float64).sum(axis=(2, 3))
        arrays[tag + "__out_absmax"] = np.float32(np.abs(y).max())
    save("g13_unet_forward.npz", **arrays)


def _params(fn):
    """[[name, kind, default-or-None as repr]] of a callable, `self` dropped"""
    import inspect
    out = []
    for prm in inspect.signature(fn).parameters.values():
        if prm.name == "self":
            continue
        out.append([prm.name, prm.kind.name, None if prm.default is inspect.Parameter.empty else repr(prm.default)])
    return out


def g14_api():
    """The reference's PUBLIC SURFACE on the hot path as data: module -> name -> call signature,
    for classes the constructor, the public methods and the public attributes of a constructed instance.  A CPU test
    checks that the product's same-named modules expose every entry call-compatibly, so that INTEGRATION.md's
    "swap the modules" statement cannot drift.  Out of scope (recorded by name only): RednerRenderer and
    OrthoToPerspectiveMapping (renderers.py:106-270), the gamma/file helpers' callers."""
    import inspect
    in_scope = {
        "renderers": ["LocalRenderer"],
        "losses": ["SVBRDFL1Loss", "RenderingLoss", "MixedLoss"],
        "environment": ["Camera", "Light", "Scene", "generate_random_scenes", "generate_specular_scenes"],
        "utils": ["pack_svbrdf", "unpack_svbrdf", "decode_svbrdf", "encode_as_unit_interval", "decode_from_unit_interval",
                  "gamma_encode", "gamma_decode", "generate_normalized_random_direction", "enable_deterministic_random_engine"],
    }
    mods = {"renderers": ref_renderers, "losses": ref_losses, "environment": ref_env, "utils": ref_utils}
    ctor_args = {"LocalRenderer": (), "SVBRDFL1Loss": (), "Camera": ([0.0, 0.0, 1.0],),
                 "Light": ([0.0, 0.0, 1.0], [1.0, 1.0, 1.0])}
    api = {}
    for mname, names in in_scope.items():
        mod, entry = mods[mname], {}
        for name in names:
            obj = getattr(mod, name)
            if inspect.isclass(obj):
                methods = {k: _params(v) for k, v in vars(obj).items()
                           if inspect.isfunction(v) and not k.startswith("_")}
                if name in ("RenderingLoss", "MixedLoss"):
                    inst = obj(ref_renderers.LocalRenderer())
                elif name == "Scene":
                    inst = obj(ref_env.Camera([0.0, 0.0, 1.0]), ref_env.Light([0.0, 0.0, 1.0], [1.0, 1.0, 1.0]))
                else:
                    inst = obj(*ctor_args[name])
                attrs = {}
                for k, v in vars(inst).items():
                    if k.startswith("_") or k == "training":
                        continue
                    attrs[k] = v if isinstance(v, (int, float)) and not isinstance(v, bool) else type(v).__name__
                if isinstance(inst, torch.nn.Module):
                    for k, v in inst._modules.items():
                        attrs[k] = type(v).__name__
                entry[name] = {"kind": "class", "init": _params(obj.__init__), "methods": methods, "attributes": attrs,
                               "is_nn_module": isinstance(inst, torch.nn.Module)}
                if name == "LocalRenderer":
                    # only `render` is called from outside renderers.py (grep of the reference: losses.py:39-40,
                    # dataset.py:212, the notebooks); the per-term helpers are the body of render() -- rows a1-a9,
                    # fused into the kernels -- and are recorded by name only
                    entry[name]["methods"] = {"render": methods["render"]}
                    entry[name]["methods_fused_into_the_kernels"] = sorted(k for k in methods if k != "render")
            else:
                entry[name] = {"kind": "function", "params": _params(obj)}
        public = sorted(k for k, v in vars(mod).items() if not k.startswith("_") and getattr(v, "__module__", None) == mod.__name__)
        api[mname] = {"in_scope": entry, "out_of_scope_names": [k for k in public if k not in names]}
    # identity of the code the kernels restate (renderers.py:8-104): fingerprints of dot_product, normalize and the
    # nine methods of LocalRenderer, computed by the product's own function from the imported reference.  The product
    # takes the fused kernel for a foreign `renderers.LocalRenderer` object only when its class matches them.
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from svbrdf_estimation_amd import _refcode
    fp = _refcode.fingerprint(ref_renderers.LocalRenderer)
    api["renderers"]["code_identity"] = {"functions": list(_refcode.MODULE_FUNCTIONS) + list(_refcode.METHODS),
                                         "source": fp["source"], "bytecode": {fp["python"]: fp["bytecode"]}}
    with open(os.path.join(HERE, "g14_api.json"), "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)
    print("wrote g14_api.json")


def _double_maps(seed, B, H, tiled=True):
    """float64 maps that are NOT float32-representable: synth maps plus a 2^-30-scale perturbation, normals renormalised"""
    m = synth.make_maps(seed, B, H, tiled_roughness=tiled).astype(np.float64)
    jit = (synth.uniform01(seed * 7 + 3, m.shape).astype(np.float64) - 0.5) * 2.0 ** -29
    m = m + jit
    n = m[:, 0:3]
    m[:, 0:3] = n / np.sqrt((n ** 2).sum(axis=1, keepdims=True))
    m[:, 3:] = np.clip(m[:, 3:], 0.0, 1.0)
    return m


def g15_float64():
    """Row b: LocalRenderer.render and RenderingLoss are dtype-agnostic in the reference (renderers.py:67-104,
    losses.py:29-52).  With float64 maps the reference computes in mixed precision -- grid, positions and colours are
    float32 (torch.linspace default dtype, torch.Tensor(...)), everything touching the maps is promoted to double.
    Frozen here: renders of [2,12,16,16] double maps under a random, a specular and a grazing scene with the gradient
    of a double cotangent, a roughness-below-clamp / back-facing edge patch, and RenderingLoss / MixedLoss with their
    gradients on double inputs (scenes recorded)."""
    out = {}
    B, H = 2, 16
    maps = _double_maps(151, B, H, tiled=False)
    maps[0, 6:9, :2, :] = 0.0004                                     # roughness below the clamp (zero gradient there)
    maps[1, 0:3, 3, :] = np.array([0.8, 0.0, -0.6])[:, None]         # back-facing normals (n.wi < 0 for frontal lights)
    torch.manual_seed(21)
    scenes = ref_env.generate_random_scenes(1) + ref_env.generate_specular_scenes(1) + [
        ref_env.Scene(ref_env.Camera([1.5, -0.4, 0.05]), ref_env.Light([-0.7, 0.9, 0.6], [30.0, 20.0, 10.0]))]
    R = ref_renderers.LocalRenderer()
    x = torch.from_numpy(maps).clone().requires_grad_(True)
    rend = torch.stack([R.render(sc, x) for sc in scenes], dim=1)     # [B,S,3,H,W]
    assert rend.dtype == torch.float64
    cot = (synth.uniform01(991, tuple(rend.shape)).astype(np.float64) - 0.5)
    (rend * torch.from_numpy(cot)).sum().backward()
    out.update(render_maps=maps, render_scenes=scene_table(scenes), render_out=rend.detach().numpy(), render_cot=cot,
               render_grad=x.grad.numpy())
    inp, tgt = _double_maps(161, B, H), _double_maps(162, B, H)
    for name, fn in (("loss", ref_losses.RenderingLoss(ref_renderers.LocalRenderer())),
                     ("mixed", ref_losses.MixedLoss(ref_renderers.LocalRenderer()))):
        x = torch.from_numpy(inp).clone().requires_grad_(True)
        torch.manual_seed(33)
        with _Recorder() as rec:
            val = fn(x, torch.from_numpy(tgt))
        assert val.dtype == torch.float64
        val.backward()
        out.update({name + "_value": np.float64(val.item()), name + "_grad": x.grad.numpy(), name + "_scenes": rec.table()})
    out.update(loss_input=inp, loss_target=tgt, loss_rng_seed=np.int64(33))
    save("g15_float64.npz", **out)


def g16_second_order():
    """Row b, second order: the reference's render and losses are built from differentiable torch ops, so
    ``backward(create_graph=True)`` / ``torch.autograd.grad(..., create_graph=True)`` work through them (renderers.py:67-104,
    losses.py:29-52).  Frozen here, all in float64 (the reference's mixed precision for double maps):
      * render: maps [2,12,12,12] with the edge rows of g15, three scenes, a cotangent c and a direction v ->
        g = d<c, render(x)>/dx, then d<g, v>/dx (Hessian-vector product) and d<g, v>/dc (= J v);
      * RenderingLoss and MixedLoss on double inputs: g = dL/dx, the gradient of the penalty sum(g^2) w.r.t. x, and the
        Hessian-vector product d<g, v>/dx (scenes recorded);
      * the same two losses on float32-VALUED inputs evaluated in double ("f32v_*"): what a float32 caller's
        create_graph=True is compared with (the engine promotes such a call to double)."""
    out = {}
    B, H = 2, 12
    maps = _double_maps(171, B, H, tiled=False)
    maps[0, 6:9, :2, :] = 0.0004                                     # roughness below the clamp
    maps[1, 0:3, 3, :] = np.array([0.8, 0.0, -0.6])[:, None]         # back-facing normals
    torch.manual_seed(23)
    scenes = ref_env.generate_random_scenes(1) + ref_env.generate_specular_scenes(1) + [
        ref_env.Scene(ref_env.Camera([1.5, -0.4, 0.05]), ref_env.Light([-0.7, 0.9, 0.6], [30.0, 20.0, 10.0]))]
    R = ref_renderers.LocalRenderer()
    x = torch.from_numpy(maps).clone().requires_grad_(True)
    rend = torch.stack([R.render(sc, x) for sc in scenes], dim=1)     # [B,S,3,H,W]
    cot = torch.from_numpy(synth.uniform01(993, tuple(rend.shape)).astype(np.float64) - 0.5).requires_grad_(True)
    v = synth.uniform01(995, maps.shape).astype(np.float64) - 0.5
    (g,) = torch.autograd.grad((rend * cot).sum(), x, create_graph=True)
    hv, jv = torch.autograd.grad((g * torch.from_numpy(v)).sum(), (x, cot))
    out.update(render_maps=maps, render_scenes=scene_table(scenes), render_cot=cot.detach().numpy(), render_v=v,
               render_grad=g.detach().numpy(), render_hvp=hv.numpy(), render_jv=jv.numpy())
    inp, tgt = _double_maps(181, B, H), _double_maps(182, B, H)
    inp32 = synth.make_maps(183, B, H).astype(np.float64)            # float32-valued
    tgt32 = synth.make_maps(184, B, H).astype(np.float64)
    vl = synth.uniform01(997, inp.shape).astype(np.float64) - 0.5
    for tag, a, b in (("", inp, tgt), ("f32v_", inp32, tgt32)):
        for name, fn in (("loss", ref_losses.RenderingLoss(ref_renderers.LocalRenderer())),
                         ("mixed", ref_losses.MixedLoss(ref_renderers.LocalRenderer()))):
            x = torch.from_numpy(a).clone().requires_grad_(True)
            torch.manual_seed(35)
            with _Recorder() as rec:
                val = fn(x, torch.from_numpy(b))
            (g,) = torch.autograd.grad(val, x, create_graph=True)
            (pen,) = torch.autograd.grad((g ** 2).sum(), x, retain_graph=True)
            (hv,) = torch.autograd.grad((g * torch.from_numpy(vl)).sum(), x)
            out.update({tag + name + "_value": np.float64(val.item()), tag + name + "_grad": g.detach().numpy(),
                        tag + name + "_penalty_grad": pen.numpy(), tag + name + "_hvp": hv.numpy(),
                        tag + name + "_scenes": rec.table()})
    out.update(loss_input=inp, loss_target=tgt, f32v_loss_input=inp32, f32v_loss_target=tgt32, loss_v=vl,
               loss_rng_seed=np.int64(35))
    save("g16_second_order.npz", **out)


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
    if len(sys.argv) > 1 and sys.argv[1] == "--only-dataset":        # row f4 reader + mix, added in round 2
        g12_dataset_reader()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-api":            # row b public surface, added in round 3
        g14_api()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-float64":        # row b dtype-agnostic render / loss, added in round 4
        g15_float64()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-second-order":   # row b create_graph=True, added in round 4
        g16_second_order()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-unet":           # row f4 network forward, added in round 2
        g13_unet_forward()
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
    g12_dataset_reader()
    g13_unet_forward()
    g14_api()
    g15_float64()
    g16_second_order()
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
