"""CPU: host-side logic of the drop-in modules (scene sampler, SVBRDF packing helpers, error
behaviour) and the C ABI's load/export contract.  No kernel is launched here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _intel_host():
    try:
        return "GenuineIntel" in open("/proc/cpuinfo").read()
    except OSError:
        return False


def assert_same_as_fixture(a, b, what=""):
    """Bit-identical on the kind of host the fixtures were generated on (Intel: same MKL code
    path as the reference run); elsewhere torch's CPU sqrt/cos/exp go through a different MKL
    code path and may differ in the last bit -- measured on the EPYC GPU box: 1 ULP."""
    a, b = np.asarray(a), np.asarray(b)
    if _intel_host():
        assert np.array_equal(a, b), what
    else:
        np.testing.assert_allclose(a, b, rtol=3e-7, atol=1e-7, err_msg=what)


# ------------------------------------------------------------------ C ABI

def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "svbrdf_hip.h")).read()
    return re.findall(r"SVBRDF_API\s+[\w\s\*]+?\b(svbrdf_\w+)\s*\(", text)


def test_library_loads_and_exports_every_declared_symbol():
    from svbrdf_estimation_amd import _native
    lib = _native._load()                     # dlopen + ABI version check, no GPU needed
    names = _declared_symbols()
    assert len(names) >= 9 and "svbrdf_rendering_loss_fwd_bwd" in names
    for n in names:
        assert hasattr(lib, n), "libsvbrdf_hip.so does not export %s" % n
    assert lib.svbrdf_abi_version() == _native.ABI_VERSION


def test_make_xrow_matches_reference_linspace_bits(golden):
    from svbrdf_estimation_amd import _native
    g = golden("g6_linspace.npz")
    for k in g.files:
        W = int(k[2:])
        assert np.array_equal(_native.make_xrow_host(W).numpy().view(np.uint32), g[k].view(np.uint32)), W
        assert torch.equal(_native.make_xrow_host(W), torch.linspace(-1, 1, W))


def test_argument_errors_without_gpu():
    from svbrdf_estimation_amd import _native
    lib = _native._load()
    assert lib.svbrdf_render_fwd(None, None, None, None, 1, 1, 4, 4, None) == -1
    assert b"null" in lib.svbrdf_last_error()
    assert lib.svbrdf_make_xrow(None, 4) == -1
    assert lib.svbrdf_rendering_loss_workspace_bytes(8, 9, 256, 256) == 65 * 8
    assert lib.svbrdf_rendering_loss_workspace_bytes(0, 9, 256, 256) == 0


def test_no_cpu_fallback(monkeypatch):
    """no GPU => the product path fails loudly instead of computing somewhere else.  (With a ROCm device present a HOST tensor
    handed to ``render`` -- the reference dataloader's call shape, dataset.py:206-212 -- is staged to the device and rendered
    by K1, forward only: tests/test_gpu_parity.py.  Without a GPU every call must raise, and the message of
    the renderer's must name the way out.)"""
    from svbrdf_estimation_amd import NativeLibraryError, environment, losses, renderers
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)      # the machine this test describes (true here anyway)
    sc = environment.Scene(environment.Camera([0, 0, 2.0]), environment.Light([0, 0, 2.0], [1.0, 1.0, 1.0]))
    with pytest.raises(NativeLibraryError, match="patch_renderer=False"):
        renderers.LocalRenderer().render(sc, torch.zeros(12, 8, 8))
    with pytest.raises(NativeLibraryError, match="patch_renderer=False"):
        renderers.LocalRenderer().render(sc, torch.zeros(2, 12, 8, 8, dtype=torch.float64))
    with pytest.raises(NativeLibraryError, match="forward-only"):       # refused before any device is looked for
        renderers.LocalRenderer().render(sc, torch.zeros(12, 8, 8, requires_grad=True))
    with pytest.raises(NativeLibraryError):
        losses.RenderingLoss(renderers.LocalRenderer())(torch.zeros(1, 12, 8, 8), torch.zeros(1, 12, 8, 8))
    with pytest.raises(ValueError):
        losses.RenderingLoss(renderers.LocalRenderer())(torch.zeros(12, 8, 8), torch.zeros(12, 8, 8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "svbrdf_estimation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "__init__.py" and "oracle" not in text, (
                    "%s mentions the oracle" % os.path.join(dirpath, f))


# ------------------------------------------------------------------ scene sampler (environment.py)

def test_scene_sampler_bit_exact_vs_reference(golden):
    from svbrdf_estimation_amd import environment as env, utils
    g = golden("g5_scene_sampler.npz")
    for seed in (0, 7, 313):
        torch.manual_seed(seed)
        assert_same_as_fixture(env.scene_table(3, 6).numpy(), g["seed_%d" % seed], seed)
    torch.manual_seed(5)
    assert_same_as_fixture(env.scene_table(11, 21).numpy(), g["seed_5_11_21"])
    torch.manual_seed(3)
    assert_same_as_fixture(utils.generate_normalized_random_direction(8, 0.001, 0.05).numpy(), g["seed_3_dirs_8"])
    # SURVEY.md section 4 known answers (seed 7)
    torch.manual_seed(7)
    scenes = env.generate_random_scenes(3) + env.generate_specular_scenes(6)
    np.testing.assert_allclose(scenes[0].camera.pos.numpy(), [-0.38334444, -0.57874203, 0.71979487], rtol=1e-6)
    np.testing.assert_allclose(scenes[3].light.pos.numpy(), [-4.70324516, 0.05648994, 8.91850853], rtol=1e-6)
    assert scenes[0].light.color == [20.0, 20.0, 20.0] and scenes[3].light.color == [50.0, 50.0, 50.0]
    assert_same_as_fixture(env.scene_to_row(scenes[3]).numpy(), g["seed_7"][3])


@pytest.mark.parametrize("B,R,M", [(8, 3, 6), (2, 11, 21), (1, 3, 6), (5, 0, 4), (3, 2, 0), (2, 17, 40), (3, 1, 7), (3, 1, 8)])
def test_batch_sampler_equals_per_item_draws_and_rng_state(B, R, M):
    from svbrdf_estimation_amd import environment as env
    samp = env.BatchSceneSampler(B, R, M)
    for seed in range(25):
        torch.manual_seed(seed)
        a = torch.stack([env.scene_table(R, M) for _ in range(B)])
        sa = torch.get_rng_state()
        torch.manual_seed(seed)
        b = samp.sample()
        assert torch.equal(a, b) and torch.equal(sa, torch.get_rng_state()), (B, R, M, seed)


def test_cpp_sampler_equals_per_item_draws_and_rng_state():
    """the native host path's sampler (csrc/host_ext.cpp) against the per-item reference order"""
    from svbrdf_estimation_amd import _hostext, environment as env
    ext = _hostext.module()
    assert ext is not None, "host extension not built"
    for (B, R, M) in [(8, 3, 6), (2, 11, 21), (5, 0, 4), (3, 2, 0), (3, 1, 7), (3, 1, 8)]:
        for seed in range(10):
            torch.manual_seed(seed)
            a = torch.stack([env.scene_table(R, M) for _ in range(B)])
            sa = torch.get_rng_state()
            torch.manual_seed(seed)
            b = ext.sample_scene_table(B, R, M)
            assert torch.equal(a, b) and torch.equal(sa, torch.get_rng_state()), (B, R, M, seed)


def test_cpp_input_scene_sampler_equals_per_sample_draws_and_rng_state(golden):
    """row f3: the native sampler of the input-photo scenes (csrc/host_ext.cpp sample_input_scene_table) against the
    per-sample Python restatement of dataset.py:172-204 -- tables and generator state bit for bit, for every shape class
    (one photo: no hemisphere draws; fewer and more than 16 normal draws per call: ATen's two code paths; with and without
    augmentation) -- and against the scenes the reference itself produced (g10)."""
    from svbrdf_estimation_amd import _hostext, synthesis
    ext = _hostext.module()
    assert ext is not None, "host extension not built"
    for aug in (False, True):
        for n in (1, 2, 4, 5, 6, 16, 20):              # 3n >= 16 from n = 6 (white balance), n >= 16 (light power)
            for B in (1, 3, 8):
                for seed in range(4):
                    torch.manual_seed(seed)
                    a = torch.stack([synthesis.input_scene_table(n, aug) for _ in range(B)])
                    sa = torch.get_rng_state()
                    torch.manual_seed(seed)
                    b = ext.sample_input_scene_table(B, n, aug)
                    assert torch.equal(a, b) and torch.equal(sa, torch.get_rng_state()), (aug, n, B, seed)
    g = golden("g10_render_inputs.npz")
    for aug in (0, 1):
        for n in (1, 4):
            k = "aug%d_n%d" % (aug, n)
            torch.manual_seed(int(g[k + "__seed"]))
            assert_same_as_fixture(ext.sample_input_scene_table(1, n, bool(aug))[0].numpy(), g[k + "__scenes"], k)
    with pytest.raises(RuntimeError):
        ext.sample_input_scene_table(0, 1, True)


def test_rendering_loss_sampling_order_matches_reference(golden):
    from svbrdf_estimation_amd import losses, renderers
    g = golden("g5_scene_sampler.npz")
    fn = losses.RenderingLoss(renderers.LocalRenderer())
    assert (fn.random_configuration_count, fn.specular_configuration_count) == (3, 6)
    torch.manual_seed(99)
    assert_same_as_fixture(fn.sample_scene_table(2).numpy(), g["seed_99_two_items"])
    g3 = golden("g3_loss_7_s5.npz")
    fn.random_configuration_count, fn.specular_configuration_count = int(g3["n_random"]), int(g3["n_specular"])
    torch.manual_seed(int(g3["rng_seed"]))
    assert_same_as_fixture(fn.sample_scene_table(3).numpy(), g3["scenes"])


def test_input_synthesis_scene_tables_bit_exact(golden):
    """row f3: scene construction of dataset.py:172-204 (with and without augmentation)"""
    from svbrdf_estimation_amd import synthesis
    g = golden("g10_render_inputs.npz")
    for aug in (0, 1):
        for n in (1, 4):
            k = "aug%d_n%d" % (aug, n)
            torch.manual_seed(int(g[k + "__seed"]))
            assert_same_as_fixture(synthesis.input_scene_table(n, bool(aug)).numpy(), g[k + "__scenes"], k)
    with pytest.raises(ValueError):
        synthesis.render_inputs(torch.zeros(9, 4, 4), 1)
    with pytest.raises(Exception):
        synthesis.render_inputs(torch.zeros(12, 4, 4), 1)      # CPU tensor: no fallback


def test_noise_field_reference_generator_matches_the_published_philox_vectors():
    """tests/philox_ref.py is the checker of the fused sensor-noise epilogue (svbrdf_render_inputs, GPU suite); its
    Philox4x32-10 is pinned by the known-answer vectors the Random123 distribution ships for philox4x32 with 10 rounds
    (counter, key -> output: zeros, all ones, digits of pi)."""
    import philox_ref
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(v) for v in philox_ref.philox4x32_10(*ctr, *key))
        assert got == want, (ctr, key, [hex(v) for v in got])
    # vectorised over counters = scalar calls; the normal field is standard normal and a function of (seed, offset, index)
    c0 = np.arange(5, dtype=np.uint64)
    vec = philox_ref.philox4x32_10(c0, 0, 7, 1, 3, 9)
    for i in range(5):
        assert tuple(int(v[i]) for v in vec) == tuple(int(v) for v in philox_ref.philox4x32_10(i, 0, 7, 1, 3, 9))
    n = philox_ref.normal_field(1234, 8, 1 << 18)
    assert abs(n.mean()) < 4.0 / np.sqrt(n.size) and abs(n.var() - 1.0) < 4.0 * np.sqrt(2.0 / n.size)
    assert abs((n ** 4).mean() - 3.0) < 0.1 and np.abs(n).max() < 6.0
    assert np.array_equal(n[:1000], philox_ref.normal_field(1234, 8, 1000))
    assert not np.array_equal(n[:1000], philox_ref.normal_field(1234, 12, 1000))
    assert abs(np.corrcoef(n[:-1], n[1:])[0, 1]) < 0.01 and abs(np.corrcoef(n[:-4], n[4:])[0, 1]) < 0.01


# ------------------------------------------------------------------ utils.py

def test_utils_against_reference(golden):
    from svbrdf_estimation_amd import utils
    g = golden("g8_utils.npz")
    x = torch.from_numpy(g["enc9"]).requires_grad_(True)
    dec = utils.decode_svbrdf(x)
    assert_same_as_fixture(dec.detach().numpy(), g["decoded12"])
    dec.backward(torch.from_numpy(g["cot"]))
    np.testing.assert_allclose(x.grad.numpy(), g["grad9"], rtol=1e-6, atol=1e-7)
    assert_same_as_fixture(utils.decode_svbrdf(torch.from_numpy(g["enc9"][0])).numpy(), g["decoded12_single"])
    img = torch.from_numpy(g["img"])
    assert_same_as_fixture(utils.gamma_encode(img).numpy(), g["gamma_enc"])
    assert_same_as_fixture(utils.gamma_decode(img).numpy(), g["gamma_dec"])
    assert np.array_equal(utils.encode_as_unit_interval(torch.from_numpy(g["enc9"])).numpy(), g["unit"])
    assert np.array_equal(utils.decode_from_unit_interval(img).numpy(), g["from_unit"])
    n, d, r, s = utils.unpack_svbrdf(torch.from_numpy(g["maps404"]))
    for a, k in ((n, "n"), (d, "d"), (r, "r"), (s, "s")):
        assert np.array_equal(a.numpy(), g[k])
    assert np.array_equal(utils.pack_svbrdf(n, d, r, s).numpy(), g["repacked"])
    with pytest.raises(ValueError):
        utils.unpack_svbrdf(torch.zeros(9, 4, 4))


def test_reference_unit_test_constants():
    """the reference's own unit tests (utils.py:149-247): gamma magic pixel, channel order"""
    from svbrdf_estimation_amd import utils
    enc = torch.tensor([[[1.3703509847201]], [[1.3703509847201]]])
    torch.testing.assert_close(utils.gamma_decode(enc), torch.full_like(enc, 2.0))
    torch.testing.assert_close(utils.gamma_encode(torch.full_like(enc, 2.0)), enc)
    torch.testing.assert_close(utils.gamma_decode(enc.unsqueeze(0).repeat(5, 1, 1, 1)), torch.full((5, 2, 1, 1), 2.0))
    nv = 1.0 / 3.0 ** 0.5
    n = torch.full((3, 1, 1), nv)
    d = torch.tensor([0.1, 0.2, 0.3]).view(3, 1, 1)
    r = torch.full((3, 1, 1), 0.3)
    s = torch.tensor([0.4, 0.5, 0.6]).view(3, 1, 1)
    sv = utils.pack_svbrdf(n, d, r, s)
    assert tuple(sv.shape) == (12, 1, 1)
    assert torch.equal(sv[0:3], n) and torch.equal(sv[3:6], d) and torch.equal(sv[6:9], r) and torch.equal(sv[9:12], s)
    batch = sv.repeat(5, 1, 1, 1)
    bn, bd, br, bs = utils.unpack_svbrdf(batch)
    assert tuple(bd.shape) == (5, 3, 1, 1) and torch.equal(bs[2], s) and torch.equal(bn[4], n)


def test_l1_and_plugin_losses_on_cpu(golden):
    """SVBRDFL1Loss is stock torch and the plugin path works with any foreign renderer on any device"""
    from svbrdf_estimation_amd import losses
    g = golden("g3_loss_48.npz")
    x = torch.from_numpy(g["input"]).requires_grad_(True)
    l1 = losses.SVBRDFL1Loss()(x, torch.from_numpy(g["target"]))
    l1.backward()
    assert abs(l1.item() - float(g["l1_loss"])) <= 1e-6 * float(g["l1_loss"])
    np.testing.assert_allclose(x.grad.numpy(), g["l1_grad"], rtol=1e-5, atol=1e-9)

    class Flat:
        calls = 0

        def render(self, scene, svbrdf):
            Flat.calls += 1
            return svbrdf[3:6].unsqueeze(0) * float(scene.light.color[0])

    fn = losses.RenderingLoss(Flat())
    fn.random_configuration_count, fn.specular_configuration_count = 1, 2
    torch.manual_seed(0)
    loss = fn(torch.rand(2, 12, 4, 4), torch.rand(2, 12, 4, 4))
    assert loss.dim() == 0 and Flat.calls == 2 * 2 * 3
    mixed = losses.MixedLoss(Flat(), l1_weight=0.25)
    assert mixed.l1_weight == 0.25 and isinstance(mixed.rendering_loss, losses.RenderingLoss)


def test_shard_helpers():
    from svbrdf_estimation_amd import distributed as D
    assert [D.shard_bounds(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        D.shard_bounds(10, 0, 4)
    t = torch.arange(12).view(6, 2)
    assert torch.equal(D.shard(t, 1, 3), t[2:4])
    assert D.rank_seed(313, 5) == 318
    assert torch.equal(D.global_mean(torch.tensor(2.0)), torch.tensor(2.0))   # no process group: identity


def test_committed_bench_line_follows_the_contract():
    """profiles/r<NN>_bench.json is a bench.py output line: the keys the driver and the judge read must be there"""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    for path in files:
        with open(path) as f:
            j = json.load(f)
        for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                         ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str),
                         ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
            assert isinstance(j[key], typ), (path, key)
        assert j["vs_baseline"] is None and j["scaling"] == "weak" and j["unit"] == "patches/s" and j["dtype"] == "f32"
        assert "workload" in j["config"] and "model" not in j["config"]
        r = j["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
        assert r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"]
        c = j["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == j["unit"] and c["sample"]
        # the number is consistent with its own step time
        assert abs(j["value"] - j["config"]["global_batch"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


def test_header_is_valid_c_and_the_c_host_builds(tmp_path):
    """include/svbrdf_hip.h must be consumable by a C compiler (the boundary is a C ABI): compiled alone as strict
    C99, every declared function referenced; and the plain-C host program of the GPU suite must build and link"""
    import re
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include/hip"):
        pytest.skip("toolchain test: needs gcc and the ROCm headers under /opt/rocm/include")
    header = os.path.join(ROOT, "include", "svbrdf_hip.h")
    names = sorted(set(re.findall(r"\b(svbrdf_\w+)\s*\(", open(header).read())))
    assert len(names) >= 17
    src = tmp_path / "use_header.c"
    src.write_text('#include "svbrdf_hip.h"\n#include <stddef.h>\n'
                   "typedef void (*fn)(void);\n"
                   "fn table[] = {%s};\nint main(void) { return table[0] == NULL; }\n"
                   % ", ".join("(fn)%s" % n for n in names))
    lib = os.path.join(ROOT, "svbrdf_estimation_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-Wno-pedantic-ms-format",
                           "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "use_header"),
                           "-L" + lib, "-lsvbrdf_hip", "-Wl,-rpath," + lib, "-Wl,--allow-shlib-undefined"])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_host", "render_host.c"),
                           "-o", str(tmp_path / "render_host"), "-L" + lib, "-lsvbrdf_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])


# ---------------------------------------------------------------- row b: the reference's public surface, pinned
def _sig_params(fn):
    import inspect
    return [p for p in inspect.signature(fn).parameters.values() if p.name != "self"]


def _assert_call_compatible(ref_params, fn, what):
    """every call the reference's signature accepts must bind to `fn` the same way: same parameter names in the same
    positions, the same ones optional with equal defaults; `fn` may only ADD optional parameters after them"""
    import inspect
    got = _sig_params(fn)
    if ref_params and ref_params[0][1] == "VAR_POSITIONAL":            # reference: def __init__(self, *args, **kwargs) of
        return                                                         # object / nn.Module -- called without arguments
    assert len(got) >= len(ref_params), "%s: %d parameters, the reference has %d" % (what, len(got), len(ref_params))
    for (name, kind, default), p in zip(ref_params, got):
        assert p.name == name and p.kind.name == kind, "%s: parameter %r (%s) vs reference %r (%s)" % (what, p.name, p.kind.name, name, kind)
        if default is None:
            assert p.default is inspect.Parameter.empty, "%s: %s must stay required" % (what, name)
        else:
            assert repr(p.default) == default, "%s: default of %s is %r, reference %s" % (what, name, p.default, default)
    for p in got[len(ref_params):]:
        assert p.default is not inspect.Parameter.empty or p.kind.name in ("VAR_POSITIONAL", "VAR_KEYWORD"), \
            "%s: extra parameter %s has no default" % (what, p.name)


def test_public_surface_matches_the_reference():
    """tests/golden/g14_api.json = the reference's hot-path modules as data (names, call signatures, public attributes
    of constructed objects; make_golden.py g14_api).  INTEGRATION.md section 1 swaps the reference's flat modules for
    this package's same-named ones; this pins that every name a caller can reach through them exists here and binds
    the same calls."""
    import importlib
    import inspect
    import json
    with open(os.path.join(ROOT, "tests", "golden", "g14_api.json")) as f:
        api = json.load(f)
    assert set(api) == {"renderers", "losses", "environment", "utils"}
    from svbrdf_estimation_amd import environment, renderers
    for mname, spec in api.items():
        mod = importlib.import_module("svbrdf_estimation_amd." + mname)
        for name, ref in spec["in_scope"].items():
            assert hasattr(mod, name), "svbrdf_estimation_amd.%s lacks %s" % (mname, name)
            obj = getattr(mod, name)
            what = "%s.%s" % (mname, name)
            if ref["kind"] == "function":
                _assert_call_compatible(ref["params"], obj, what)
                continue
            assert inspect.isclass(obj), what
            _assert_call_compatible(ref["init"], obj.__init__, what + ".__init__")
            for meth, params in ref["methods"].items():
                assert callable(getattr(obj, meth, None)), "%s lacks method %s" % (what, meth)
                _assert_call_compatible(params, getattr(obj, meth), "%s.%s" % (what, meth))
            if name in ("RenderingLoss", "MixedLoss"):
                inst = obj(renderers.LocalRenderer())
            elif name == "Scene":
                inst = obj(environment.Camera([0.0, 0.0, 1.0]), environment.Light([0.0, 0.0, 1.0], [1.0, 1.0, 1.0]))
            elif name == "Camera":
                inst = obj([0.0, 0.0, 1.0])
            elif name == "Light":
                inst = obj([0.0, 0.0, 1.0], [1.0, 1.0, 1.0])
            else:
                inst = obj()
            assert isinstance(inst, torch.nn.Module) == ref["is_nn_module"], what
            for attr, val in ref["attributes"].items():
                assert hasattr(inst, attr), "%s instance lacks attribute %s" % (what, attr)
                if isinstance(val, (int, float)):
                    assert getattr(inst, attr) == val, "%s.%s = %r, reference %r" % (what, attr, getattr(inst, attr), val)
                else:
                    assert type(getattr(inst, attr)).__name__ == val, "%s.%s is a %s, reference %s" % (
                        what, attr, type(getattr(inst, attr)).__name__, val)
    # what is deliberately absent is exactly what DESIGN.md lists as out of scope
    assert set(api["renderers"]["out_of_scope_names"]) == {"OrthoToPerspectiveMapping", "RednerRenderer", "dot_product", "normalize"}


def test_reference_code_identity_is_the_recorded_one():
    """`_refcode.REFERENCE` (what the product compares a foreign `renderers.LocalRenderer` with) = the fingerprints
    tests/golden/make_golden.py recorded from the imported reference (g14_api.json, "code_identity")."""
    import json
    from svbrdf_estimation_amd import _refcode
    with open(os.path.join(ROOT, "tests", "golden", "g14_api.json")) as f:
        ident = json.load(f)["renderers"]["code_identity"]
    assert ident["functions"] == list(_refcode.MODULE_FUNCTIONS) + list(_refcode.METHODS) and len(ident["functions"]) == 11
    assert _refcode.REFERENCE["source"] == ident["source"] and len(ident["source"]) == 64
    assert _refcode.REFERENCE["bytecode"] == ident["bytecode"]


_FORK_SOURCE = '''
import torch

def dot_product(a, b):
    return torch.sum(a * b, dim=-3, keepdim=True)

def normalize(a):
    return a / torch.sqrt(dot_product(a, a))

class LocalRenderer:
    """a fork: same module name, class name, method names and signatures as the reference's -- different shading"""
    exposure = 1.5
    calls = 0
    def xi(self, x): return (x > 0.0) * torch.ones_like(x)
    def compute_diffuse_term(self, diffuse, ks): return diffuse
    def compute_microfacet_distribution(self, roughness, NH): return roughness
    def compute_fresnel(self, specular, VH): return specular
    def compute_g1(self, roughness, XH, XN): return roughness
    def compute_geometry(self, roughness, VH, LH, VN, LN): return roughness
    def compute_specular_term(self, wi, wo, normals, diffuse, roughness, specular): return specular, specular
    def evaluate_brdf(self, wi, wo, normals, diffuse, roughness, specular): return diffuse + specular
    def render(self, scene, svbrdf):
        type(self).calls += 1
        n, d, r, s = torch.split(svbrdf, (3, 3, 3, 3), dim=-3)
        light = torch.as_tensor(scene.light.pos, dtype=svbrdf.dtype).view(3, 1, 1)
        shade = torch.clamp(dot_product(normalize(n), normalize(light.expand_as(n))), min=0.0)
        return ((d + s * r) * shade * self.exposure).unsqueeze(0)      # a camera exposure: not what the kernels compute
'''


def test_a_fork_named_like_the_reference_renderer_is_a_plugin(monkeypatch):
    """losses.py:22-23 injects the renderer: a module called `renderers` with a class called `LocalRenderer` that is NOT
    the reference's code (renderers.py:102 itself says `# TODO: Add camera exposure`) must be rendered by calling it.
    Round 3 recognised the reference's class by module and class name; now by the fingerprint of its code."""
    import types
    from svbrdf_estimation_amd import _refcode, losses
    fork = types.ModuleType("renderers")
    import linecache
    fname = "<fork-of-renderers>"
    linecache.cache[fname] = (len(_FORK_SOURCE), None, _FORK_SOURCE.splitlines(True), fname)    # inspect.getsource works
    exec(compile(_FORK_SOURCE, fname, "exec"), fork.__dict__)
    monkeypatch.setitem(sys.modules, "renderers", fork)
    r = fork.LocalRenderer()
    assert type(r).__name__ == "LocalRenderer" and type(r).__module__ == "renderers" and type(r).__mro__[1:] == (object,)
    fp = _refcode.fingerprint(type(r))
    assert fp is not None and fp["source"] is not None and fp["source"] != _refcode.REFERENCE["source"]
    loss_fn = losses.RenderingLoss(r)
    assert not loss_fn.uses_fused_kernel() and not _refcode.is_reference_local_renderer(r)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 12, 8, 8, generator=g).requires_grad_(True)
    t = torch.rand(2, 12, 8, 8, generator=g)
    loss = loss_fn(x, t)                                   # CPU tensors: only the plugin loop can run this
    loss.backward()
    assert type(r).calls == 2 * 2 * 9 and torch.isfinite(loss) and x.grad is not None and x.grad.abs().sum() > 0
    # MixedLoss with such a renderer is the literal sum too
    assert torch.isfinite(losses.MixedLoss(r)(x.detach(), t))
    # this package's own renderer: a subclass or an instance that overrides render() is a plugin as well
    from svbrdf_estimation_amd import renderers as own

    class Exposed(own.LocalRenderer):
        def render(self, scene, svbrdf):
            return super().render(scene, svbrdf) * 2.0
    assert losses.RenderingLoss(own.LocalRenderer()).uses_fused_kernel() and not losses.RenderingLoss(Exposed()).uses_fused_kernel()
    patched = own.LocalRenderer()
    patched.render = lambda scene, svbrdf: None
    assert not losses.RenderingLoss(patched).uses_fused_kernel()


_REFERENCE = "/root/reference/development/multiImage_pytorch"


@pytest.mark.skipif(not os.path.isdir(_REFERENCE), reason="needs the reference checkout (build container only)")
def test_install_patches_the_reference_modules_in_a_fresh_interpreter():
    """INTEGRATION.md section 1 against the REAL reference, in a child interpreter (the reference's flat module names
    would shadow this suite's): after ``svbrdf_estimation_amd.install()`` the reference's own import lines
    (main.py:8,12; dataset.py:7,206) resolve to this engine's classes, RednerRenderer and the rest stay the
    reference's, and ``MixedLoss(LocalRenderer())`` (main.py:82-89) is wired to the fused kernel."""
    import subprocess
    code = """
import sys, types
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2")); sys.modules.setdefault("pyredner", types.ModuleType("pyredner"))
sys.path.insert(0, %r); sys.path.insert(0, %r)
import svbrdf_estimation_amd as amd
replaced = amd.install()
from losses import MixedLoss                          # main.py:8
from renderers import LocalRenderer, RednerRenderer   # main.py:12
import renderers, losses, environment, utils
assert LocalRenderer is amd.renderers.LocalRenderer and MixedLoss is amd.losses.MixedLoss
assert losses.RenderingLoss is amd.losses.RenderingLoss and renderers.LocalRenderer is LocalRenderer
assert RednerRenderer.__module__ == "renderers" and environment.__file__.startswith(%r) and utils.__file__.startswith(%r)
assert sorted(replaced) == ["losses.MixedLoss", "losses.RenderingLoss", "losses.SVBRDFL1Loss", "renderers.LocalRenderer"]
assert all(v is not None and v.__module__ in ("renderers", "losses") for v in replaced.values())
loss_function = MixedLoss(LocalRenderer())            # main.py:82-89
assert loss_function.rendering_loss.uses_fused_kernel() and loss_function.l1_weight == 0.1
import dataset                                        # dataset.py:7 `import renderers` -> :206 renderers.LocalRenderer()
assert dataset.renderers.LocalRenderer is LocalRenderer
# the reference's OWN dataloader code reaches the patched renderer with its call shape (host [1,12,H,W] maps, tensor-valued
# positions and colour, dataset.py:206-212): accepted up to the point where a GPU is needed, and on this GPU-less machine
# the error names the way out (on an MI355X the call is served by K1: tests/test_gpu_parity.py)
import torch
class FakeSelf: use_augmentation = True
try:
    dataset.SvbrdfDataset.render_inputs(FakeSelf(), torch.rand(12, 16, 16), 2)
    raise SystemExit("render_inputs computed something without a GPU")
except amd.NativeLibraryError as e:
    assert "patch_renderer=False" in str(e) and "no ROCm device" in str(e), e
# patch_renderer=False: the reference keeps ITS LocalRenderer (CPU dataloader workers), the losses still fuse
renderers.LocalRenderer = replaced["renderers.LocalRenderer"]
amd.install(patch_renderer=False)
assert renderers.LocalRenderer.__module__ == "renderers" and renderers.LocalRenderer is not amd.renderers.LocalRenderer
assert losses.MixedLoss is amd.losses.MixedLoss
assert losses.MixedLoss(renderers.LocalRenderer()).rendering_loss.uses_fused_kernel()
class Tracer(renderers.LocalRenderer):                # a subclass that may override render() is a plugin, not the kernel
    pass
assert not losses.RenderingLoss(Tracer()).uses_fused_kernel() and not losses.RenderingLoss(object()).uses_fused_kernel()
# the reference's class is recognised by its CODE: an instance with a patched render, then the class with an edited
# method (renderers.py:102 "# TODO: Add camera exposure"), stop being "the renderer the kernels restate"
ref_r = renderers.LocalRenderer()
ref_r.render = lambda scene, svbrdf: None
assert not losses.RenderingLoss(ref_r).uses_fused_kernel()
original = renderers.LocalRenderer.compute_fresnel
renderers.LocalRenderer.compute_fresnel = lambda self, specular, VH: specular
assert not losses.RenderingLoss(renderers.LocalRenderer()).uses_fused_kernel()
renderers.LocalRenderer.compute_fresnel = original
assert losses.RenderingLoss(renderers.LocalRenderer()).uses_fused_kernel()
original = renderers.normalize
renderers.normalize = lambda a: a
assert not losses.RenderingLoss(renderers.LocalRenderer()).uses_fused_kernel()
renderers.normalize = original
print("INSTALL-OK")
""" % (_REFERENCE, ROOT, _REFERENCE, _REFERENCE)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "INSTALL-OK" in r.stdout, r.stderr[-3000:]
