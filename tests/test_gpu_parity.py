"""GPU parity tests proper: the HIP kernels, called through the C ABI (ctypes) and through
the reference-shaped Python interface, against the C oracle and the golden fixtures."""
import ctypes
import os

import numpy as np
import pytest
import torch

import synth
from tolerances import (assert_grad_close, assert_loss_close, assert_render_strict,
                        assert_render_vs_reference)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (select CPU tests with -m 'not gpu')"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def native():
    from svbrdf_estimation_amd import _native
    _native._load()
    return _native


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def _np(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------- raw C ABI

def test_abi_render_fwd_bwd_raw_ctypes(dev, native, oracle):
    """calls the exported symbols directly with device pointers, as a C host would"""
    lib = native._load()
    B, S, H = 2, 3, 32
    maps = synth.make_maps(5, B, H, tiled_roughness=False)
    torch.manual_seed(1)
    from svbrdf_estimation_amd import environment
    table = torch.stack([environment.scene_table(1, 2) for _ in range(B)]).numpy()
    xrow = np.empty(H, np.float32)
    assert lib.svbrdf_make_xrow(xrow.ctypes.data_as(ctypes.c_void_p), H) == 0
    assert np.array_equal(xrow, oracle.make_xrow(H))
    d_maps, d_sc, d_x = _t(maps, dev), _t(table, dev), _t(xrow, dev)
    d_out = torch.empty(B, S, 3, H, H, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.svbrdf_render_fwd(d_maps.data_ptr(), d_sc.data_ptr(), d_x.data_ptr(), d_out.data_ptr(), B, S, H, H, st)
    assert rc == 0, lib.svbrdf_last_error()
    torch.cuda.synchronize()
    assert_render_strict(_np(d_out), oracle.render_fwd(maps, table), "raw render_fwd")
    cot = synth.uniform01(6, (B, S, 3, H, H)) - np.float32(0.5)
    d_cot, d_g = _t(cot, dev), torch.empty(B, 12, H, H, device=dev)
    rc = lib.svbrdf_render_bwd(d_maps.data_ptr(), d_sc.data_ptr(), d_x.data_ptr(), d_cot.data_ptr(), d_g.data_ptr(),
                               B, S, H, H, st)
    assert rc == 0, lib.svbrdf_last_error()
    torch.cuda.synchronize()
    assert_grad_close(_np(d_g), oracle.render_bwd(maps, table, cot), "raw render_bwd")


def test_abi_error_codes(dev, native):
    lib = native._load()
    t = torch.zeros(12 * 16, device=dev)
    p = t.data_ptr()
    assert lib.svbrdf_render_fwd(None, p, p, p, 1, 1, 4, 4, None) == -1
    assert lib.svbrdf_render_fwd(p, p, p, p, 1, 1, 4, 8, None) == -2
    assert lib.svbrdf_render_fwd(p, p, p, p, 0, 1, 4, 4, None) == -2
    assert lib.svbrdf_render_fwd(p + 2, p, p, p, 1, 1, 4, 4, None) == -3
    need = lib.svbrdf_rendering_loss_workspace_bytes(1, 1, 4, 4)
    assert need >= 8
    assert lib.svbrdf_rendering_loss_fwd_bwd(p, p, p, p, ctypes.c_float(0.1), p, p, p, need - 1, 1, 1, 4, 4, None) == -4
    assert b"workspace" in lib.svbrdf_last_error()
    for bad_eps in (0.0, 1e-12, float("nan"), 1e12):
        assert lib.svbrdf_rendering_loss_fwd_bwd(p, p, p, p, ctypes.c_float(bad_eps), p, p, p, need, 1, 1, 4, 4, None) == -2
    assert b"eps_render" in lib.svbrdf_last_error()


def test_device_division_and_sqrt_are_correctly_rounded(dev, native):
    """the shared-reciprocal division and Newton sqrt used on the ill-conditioned path must
    agree bit for bit with IEEE `/` and sqrtf (2^31 operand pairs over the working range)"""
    lib = native._load()
    lib.svbrdf_debug_check_arith.argtypes = [ctypes.c_ulonglong, ctypes.c_uint, ctypes.c_float, ctypes.c_float,
                                             ctypes.c_void_p, ctypes.c_void_p]
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    for seed, lo, hi in ((1, 1e-3, 1e3), (2, 0.05, 64.0), (3, 0.5, 4.0)):
        rc = lib.svbrdf_debug_check_arith(1 << 31, seed, lo, hi, counts.data_ptr(),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.svbrdf_last_error()
    torch.cuda.synchronize()
    assert counts.tolist() == [0, 0], "mismatches vs IEEE (div, sqrt): %s" % counts.tolist()


def test_ragged_render_entry_points(dev, native, oracle):
    """SURVEY 8b's "R (map, scene) pairs in one launch": svbrdf_render_{fwd,bwd}_ragged with a different number of
    renders per map (zero included) against the oracle map by map, and against the regular entry points when every
    map has the same count (bitwise)."""
    from svbrdf_estimation_amd import environment
    B, H = 4, 40
    counts = [2, 0, 5, 1]
    maps = synth.make_maps(31, B, H, tiled_roughness=False)
    maps[2] = synth.make_maps(32, 1, H)[0]                      # one tied-roughness map among untied ones
    torch.manual_seed(12)
    scenes = torch.cat([environment.scene_table(c, 0) if c < 3 else environment.scene_table(2, c - 2) for c in counts if c]).numpy()
    R = sum(counts)
    out = _np(native.render_fwd_ragged(_t(maps, dev), _t(scenes, dev), counts))
    cot = synth.uniform01(33, (R, 3, H, H)) - np.float32(0.5)
    grad = _np(native.render_bwd_ragged(_t(maps, dev), _t(scenes, dev), counts, _t(cot, dev)))
    r0 = 0
    for b, c in enumerate(counts):
        if c == 0:
            assert not grad[b].any()
            continue
        tab = scenes[None, r0:r0 + c]
        assert_render_strict(out[r0:r0 + c], oracle.render_fwd(maps[b:b + 1], tab)[0], "ragged fwd map %d" % b)
        assert_grad_close(grad[b:b + 1], oracle.render_bwd(maps[b:b + 1], tab, cot[None, r0:r0 + c]), "ragged bwd map %d" % b)
        r0 += c
    # uniform counts == the regular entry points, bit for bit
    S = 3
    torch.manual_seed(13)
    table = torch.stack([environment.scene_table(1, 2) for _ in range(B)])
    reg = native.render_fwd(_t(maps, dev), table.to(dev))
    rag = native.render_fwd_ragged(_t(maps, dev), table.reshape(B * S, 9).to(dev), [S] * B)
    assert torch.equal(reg.reshape(B * S, 3, H, H), rag)
    cot2 = _t(synth.uniform01(34, (B, S, 3, H, H)) - np.float32(0.5), dev)
    assert torch.equal(native.render_bwd(_t(maps, dev), table.to(dev), cot2),
                       native.render_bwd_ragged(_t(maps, dev), table.reshape(B * S, 9).to(dev), [S] * B, cot2.reshape(B * S, 3, H, H)))
    lib = native._load()
    p = reg.data_ptr()
    assert lib.svbrdf_render_fwd_ragged(p, p, None, p, p, 1, 1, 4, 4, None) == -1
    with pytest.raises(ValueError):
        native.render_fwd_ragged(_t(maps, dev), _t(scenes, dev), [1, 1, 1, 1])


def test_render_host_scene_entry_points(dev, native, oracle):
    """svbrdf_render_{fwd,bwd}_host_scenes: the scene rows in HOST memory, carried by value in the launch's argument
    block -- per-map tables and ONE table shared by every map (LocalRenderer.render's "one scene, B maps",
    renderers.py:98) -- bitwise equal to the device-table entry points and within tolerance of the oracle; raw C ABI
    call with a host pointer; the row limit."""
    from svbrdf_estimation_amd import environment
    B, S, H = 3, 4, 36
    maps = synth.make_maps(41, B, H, tiled_roughness=False)
    maps[1] = synth.make_maps(42, 1, H)[0]
    d_maps = _t(maps, dev)
    torch.manual_seed(14)
    table = torch.stack([environment.scene_table(2, 2) for _ in range(B)])          # host [B,S,9]
    cot = _t(synth.uniform01(43, (B, S, 3, H, H)) - np.float32(0.5), dev)
    dev_out, dev_grad = native.render_fwd(d_maps, table.to(dev)), native.render_bwd(d_maps, table.to(dev), cot)
    assert torch.equal(native.render_fwd(d_maps, table), dev_out)                   # host [B,S,9] by value
    assert torch.equal(native.render_bwd(d_maps, table, cot), dev_grad)
    assert_render_strict(_np(dev_out), oracle.render_fwd(maps, table.numpy()), "host-scenes fwd")
    shared = table[0]                                                               # host [S,9]: same scenes, every map
    expanded = shared.unsqueeze(0).expand(B, S, 9).contiguous().to(dev)
    assert torch.equal(native.render_fwd(d_maps, shared), native.render_fwd(d_maps, expanded))
    assert torch.equal(native.render_bwd(d_maps, shared, cot), native.render_bwd(d_maps, expanded, cot))
    # raw C ABI, host pointer
    lib = native._load()
    out = torch.empty(B, S, 3, H, H, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    host_rows = np.ascontiguousarray(shared.numpy())
    rc = lib.svbrdf_render_fwd_host_scenes(d_maps.data_ptr(), host_rows.ctypes.data_as(ctypes.c_void_p), 1,
                                           native.xrow(dev, H).data_ptr(), out.data_ptr(), B, S, H, H, st)
    assert rc == 0, lib.svbrdf_last_error()
    host_rows[:] = 0                            # the rows were copied into the launch: the caller's buffer is free again
    torch.cuda.synchronize()
    assert torch.equal(out, native.render_fwd(d_maps, expanded))
    # more rows than the argument block holds: the C ABI refuses, the binding uploads instead (same bits)
    cap = native.host_scenes_max_rows()
    p = out.data_ptr()
    assert lib.svbrdf_render_fwd_host_scenes(p, p, 0, p, p, cap + 1, 1, 4, 4, None) == -2
    assert lib.svbrdf_render_bwd_host_scenes(p, p, 1, p, p, p, 1, cap + 1, 4, 4, None) == -2
    assert lib.svbrdf_render_fwd_host_scenes(p, None, 1, p, p, 1, 1, 4, 4, None) == -1
    Sbig, Hs = cap + 7, 8
    small = _t(synth.make_maps(44, 1, Hs), dev)
    torch.manual_seed(15)
    big = environment.scene_table(Sbig, 0)
    assert torch.equal(native.render_fwd(small, big), native.render_fwd(small, big.unsqueeze(0).to(dev)))
    with pytest.raises(ValueError):
        native.rendering_loss(d_maps, d_maps, shared)                               # the loss takes [B,S,9] only


def test_c_host_program_calls_the_abi(dev, golden, tmp_path):
    """a plain-C host (tests/c_host/render_host.c: gcc, the HIP runtime's C API, include/svbrdf_hip.h and nothing else)
    renders the KAT-1 case through svbrdf_render_fwd and prints the radiance; it must be the reference's value
    (within the rendering tolerance: the kernels are a few ULP from the reference's own rounding)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "render_host")
    libdir = os.path.join(root, "svbrdf_estimation_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c_host", "render_host.c"),
                           "-o", exe, "-L" + libdir, "-lsvbrdf_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    vals = np.array([float(v) for v in out.stdout.split()[-12:]], np.float32).reshape(3, 2, 2)
    g = golden("g9_kat.npz")
    assert_render_strict(vals, g["kat1_out"][0], "C host KAT-1")


# ---------------------------------------------------------------- K1 / K2 vs oracle and goldens

@pytest.mark.parametrize("name", ["g1_render_64.npz", "g1_render_32_tiled.npz"])
def test_render_forward_golden(dev, native, oracle, golden, name):
    g = golden(name)
    maps, sc = g["maps"][None], g["scenes"][None]
    out = _np(native.render_fwd(_t(maps, dev), _t(sc, dev)))[0]
    assert_render_strict(out, oracle.render_fwd(maps, sc)[0], name + " vs oracle")
    assert_render_vs_reference(out, g["out"], oracle.render_fwd(maps, sc, f64=True)[0], name + " vs reference")


@pytest.mark.parametrize("H", [256, 512])
def test_render_forward_full_size(dev, native, oracle, golden, H):
    g = golden("g2_render_lattice_%d.npz" % H)
    maps = synth.make_maps(int(g["synth_seed"]), 1, H)
    assert synth.checksum(maps[0]) == str(g["maps_sha256"])
    sc, st = g["scenes"][None], int(g["stride"])
    out = _np(native.render_fwd(_t(maps, dev), _t(sc, dev)))[0]
    ref32, ref64 = oracle.render_fwd(maps, sc)[0], oracle.render_fwd(maps, sc, f64=True)[0]
    assert_render_strict(out, ref32, "full %d vs oracle" % H)
    assert_render_vs_reference(out[:, :, ::st, ::st], g["out_lattice"], ref64[:, :, ::st, ::st],
                               "lattice %d vs reference" % H, scale=float(g["out_max"]))
    np.testing.assert_allclose(out.astype(np.float64).sum(axis=(2, 3)), g["plane_sums"], rtol=2e-6)


def test_kat(dev, native, golden):
    g = golden("g9_kat.npz")
    for k in ("kat1", "kat2"):
        m, sc = g[k + "_maps"][None], g[k + "_scene"][None, None]
        out = _np(native.render_fwd(_t(m, dev), _t(sc, dev)))[0]
        np.testing.assert_allclose(out, g[k + "_out"], rtol=1e-6, atol=0)
        grad = _np(native.render_bwd(_t(m, dev), _t(sc, dev), torch.ones(1, 1, 3, 2, 2, device=dev)))[0]
        assert_grad_close(grad, g[k + "_grad_of_sum"], k + " grad", rtol=1e-5, afrac=1e-6)


def test_edge_cases(dev, native, oracle, golden):
    g = golden("g4_edge_cases.npz")
    for name in g["names"]:
        m, sc = g[name + "__maps"][None], g[name + "__scene"][None, None]
        out = _np(native.render_fwd(_t(m, dev), _t(sc, dev)))[0]
        ref = g[name + "__out"]
        if np.abs(ref).max() == 0:
            assert not out.any(), name
        else:
            assert_render_strict(out, oracle.render_fwd(m, sc)[0], name + " vs oracle")
            assert_render_vs_reference(out, ref, oracle.render_fwd(m, sc, f64=True)[0], name)
        grad = _np(native.render_bwd(_t(m, dev), _t(sc, dev), _t(g[name + "__cot"][None], dev)))[0]
        if np.abs(g[name + "__grad"]).max() == 0:
            assert not grad.any(), name
        else:
            assert_grad_close(grad, g[name + "__grad"], name + " grad vs reference")
            assert_grad_close(grad, oracle.render_bwd(m, sc, g[name + "__cot"][None])[0], name + " grad vs oracle")
    m, sc = g["r_below_clamp__maps"][None], g["r_below_clamp__scene"][None, None]
    grad = _np(native.render_bwd(_t(m, dev), _t(sc, dev), _t(g["r_below_clamp__cot"][None], dev)))[0]
    H = m.shape[-1]
    assert not grad[7:9, :, : H // 2].any() and not grad[6, 1:, : H // 2].any()
    assert grad[6, 0, 0] != 0


@pytest.mark.parametrize("H", [7, 17, 30, 100])
def test_ragged_sizes_all_vector_widths(dev, native, oracle, H, monkeypatch):
    """odd / non-multiple-of-4 widths force the scalar path; even ones the vector paths"""
    B, S = 2, 4
    maps = synth.make_maps(40 + H, B, H, tiled_roughness=False)
    torch.manual_seed(H)
    from svbrdf_estimation_amd import environment
    table = torch.stack([environment.scene_table(2, 2) for _ in range(B)]).numpy()
    cot = synth.uniform01(H, (B, S, 3, H, H)) - np.float32(0.5)
    ref_out = oracle.render_fwd(maps, table)
    ref_g = oracle.render_bwd(maps, table, cot)
    tgt = synth.make_maps(41 + H, B, H)
    ref_l, ref_lg = oracle.rendering_loss(maps, tgt, table)
    for vec in ("1", "2", "4"):
        for k in ("SVBRDF_K1_VEC", "SVBRDF_K2_VEC"):       # K3 is one pixel per thread by design
            monkeypatch.setenv(k, vec)
        assert_render_strict(_np(native.render_fwd(_t(maps, dev), _t(table, dev))), ref_out, "fwd H=%d vec=%s" % (H, vec))
        assert_grad_close(_np(native.render_bwd(_t(maps, dev), _t(table, dev), _t(cot, dev))), ref_g,
                          "bwd H=%d vec=%s" % (H, vec))
        l, lg = native.rendering_loss(_t(maps, dev), _t(tgt, dev), _t(table, dev))
        assert_loss_close(l.item(), ref_l, "loss H=%d vec=%s" % (H, vec))
        assert_grad_close(_np(lg), ref_lg, "loss grad H=%d vec=%s" % (H, vec))


# ---------------------------------------------------------------- K3 fused loss

@pytest.mark.parametrize("name", ["g3_loss_48.npz", "g3_loss_7_s5.npz", "g3_loss_20_untied.npz"])
def test_rendering_loss_golden(dev, native, oracle, golden, name):
    g = golden(name)
    loss, grad = native.rendering_loss(_t(g["input"], dev), _t(g["target"], dev), _t(g["scenes"], dev))
    ref_l, ref_g = oracle.rendering_loss(g["input"], g["target"], g["scenes"])
    assert_loss_close(loss.item(), ref_l, name + " vs oracle")
    assert_loss_close(loss.item(), g["loss"], name + " vs reference", rtol=2e-6)
    _, g64 = oracle.rendering_loss(g["input"], g["target"], g["scenes"], f64=True)
    assert_grad_close(_np(grad), ref_g, name + " grad vs oracle", f64=g64)
    assert_grad_close(_np(grad), g["grad_input"], name + " grad vs reference", f64=g64)
    loss_fwd, none = native.rendering_loss(_t(g["input"], dev), _t(g["target"], dev), _t(g["scenes"], dev), want_grad=False)
    assert none is None and loss_fwd.item() == loss.item()


def test_seeded_sweep_of_shapes_and_map_statistics_against_the_oracle(dev, native, oracle):
    """A seeded sweep over what the fixtures fix: batch 1-5, 1-7 scenes in every random/specular split, sizes 1-45 (every
    vector width, one-wave and multi-workgroup grids), tied and untied roughness, steep normal tilts, roughness and
    specular at and beyond the ends of their ranges (0, 1, below the 1e-3 clamp).  K1, K2, the rendering loss and the
    mixed loss through the C ABI against the oracle at the strict bounds; ties and the reference's own fp32 error are
    bounded with the oracle's fp64 instantiation as everywhere else."""
    from svbrdf_estimation_amd import environment
    rng = np.random.RandomState(20260)
    cases = 0
    for trial in range(36):
        B, H = int(rng.randint(1, 6)), int(rng.choice([1, 2, 3, 5, 8, 13, 16, 21, 31, 32, 45]))
        n_random, n_specular = int(rng.randint(0, 4)), int(rng.randint(0, 5))
        if n_random + n_specular == 0:
            n_specular = 1
        tied = bool(rng.randint(0, 2))
        tilt = float(rng.choice([0.0, 0.3, 0.9, 2.0]))
        r_lo, r_hi = [(0.0, 1.0), (0.0, 0.01), (0.2, 0.9), (0.95, 1.0)][int(rng.randint(0, 4))]
        inp = synth.make_maps(9000 + trial, B, H, tilt=tilt, r_lo=r_lo, r_hi=r_hi, tiled_roughness=tied)
        tgt = synth.make_maps(9500 + trial, B, H, tilt=0.3, tiled_roughness=bool(rng.randint(0, 2)))
        if trial % 5 == 0:      # the ends of the ranges, exactly
            inp[:, 9:12, : (H + 1) // 2] = np.float32(1.0)
            inp[:, 9:12, (H + 1) // 2:] = np.float32(0.0)
            inp[:, 3:6, :, : (H + 1) // 2] = np.float32(0.0)
        torch.manual_seed(400 + trial)
        table = torch.stack([environment.scene_table(n_random, n_specular) for _ in range(B)]).numpy()
        S = n_random + n_specular
        what = "sweep %d (B=%d S=%d+%d H=%d tied=%d tilt=%.1f r=[%.2f,%.2f])" % (trial, B, n_random, n_specular, H, tied, tilt, r_lo, r_hi)
        d_in, d_tg, d_sc = _t(inp, dev), _t(tgt, dev), _t(table, dev)
        assert_render_strict(_np(native.render_fwd(d_in, d_sc)), oracle.render_fwd(inp, table), what + " K1")
        cot = synth.uniform01(9900 + trial, (B, S, 3, H, H)) - np.float32(0.5)
        assert_grad_close(_np(native.render_bwd(d_in, d_sc, _t(cot, dev))), oracle.render_bwd(inp, table, cot), what + " K2",
                          f64=oracle.render_bwd(inp, table, cot, f64=True))
        tie = oracle.loss_tie_map(inp, tgt, table)
        for l1w, ofn in ((0.0, lambda **k: oracle.rendering_loss(inp, tgt, table, **k)),
                         (0.1, lambda **k: oracle.mixed_loss(inp, tgt, table, **k))):
            ref_l, ref_g = ofn()
            _, g64 = ofn(f64=True)
            loss, grad = native.rendering_loss(d_in, d_tg, torch.from_numpy(table) if trial % 2 else d_sc, l1_weight=l1w)   # host / device table
            assert_loss_close(loss.item(), ref_l, what)
            # (the exact-end-of-range cases render input and target equally dark at a few dozen pixels: exact ties)
            assert_grad_close(_np(grad), ref_g, what + " loss grad l1=%.1f" % l1w, f64=g64, tie_map=tie, max_ties=48)
        cases += 1
    assert cases == 36


def test_extreme_shapes_at_the_limits_of_the_abi(dev, native, oracle):
    """Maximum sizes (SURVEY 8c asks for them): the largest batch a launch takes (65,535 = grid.y; one more is refused),
    more scenes per item than the kernel-argument block or the LDS stage of the forward-only kernels hold (1,706 scenes is
    the stage's capacity: exact; 1,707 refused forward-only, fine with the gradient, whose kernel reads the table from
    memory), and a plane beyond the 32-bit plane addressing of the fused loss (refused before anything is launched)."""
    from svbrdf_estimation_amd import environment
    lib = native._load()
    # ---- B = 65,535 items of 2x2 pixels, one scene each (device table: 65,535 rows), K1 / K2 / K3 against the oracle
    B, H = 65535, 2
    inp, tgt = synth.make_maps(7101, B, H), synth.make_maps(7102, B, H)
    torch.manual_seed(71)
    one = environment.scene_table(1, 2).numpy()
    table = np.ascontiguousarray(np.tile(one[None, :1], (B, 1, 1)) + (np.arange(B, dtype=np.float32) % 7)[:, None, None] * np.float32(0.01))
    d_in, d_tg, d_sc = _t(inp, dev), _t(tgt, dev), _t(table, dev)
    assert_render_strict(_np(native.render_fwd(d_in, d_sc)), oracle.render_fwd(inp, table), "B=65535 K1")
    cot = synth.uniform01(7103, (B, 1, 3, H, H)) - np.float32(0.5)
    assert_grad_close(_np(native.render_bwd(d_in, d_sc, _t(cot, dev))), oracle.render_bwd(inp, table, cot), "B=65535 K2",
                      f64=oracle.render_bwd(inp, table, cot, f64=True))
    ref_l, ref_g = oracle.rendering_loss(inp, tgt, table)
    loss, grad = native.rendering_loss(d_in, d_tg, d_sc)
    assert_loss_close(loss.item(), ref_l, "B=65535 loss")
    assert_grad_close(_np(grad), ref_g, "B=65535 loss grad", f64=oracle.rendering_loss(inp, tgt, table, f64=True)[1],
                      tie_map=oracle.loss_tie_map(inp, tgt, table), max_ties=48)
    p = d_in.data_ptr()
    assert lib.svbrdf_render_fwd(p, p, p, p, 65536, 1, 2, 2, None) == -2 and b"65535" in lib.svbrdf_last_error()
    del d_in, d_tg, d_sc, grad
    # ---- S = 1,706 / 1,707 scenes for one item of 4x4 pixels
    H = 4
    inp, tgt = synth.make_maps(7111, 1, H), synth.make_maps(7112, 1, H)
    torch.manual_seed(72)
    big = environment.scene_table(569, 1138).numpy()[None]                # 1,707 rows
    d_in, d_tg = _t(inp, dev), _t(tgt, dev)
    for S, forward_only_ok in ((1706, True), (1707, False)):
        tab = np.ascontiguousarray(big[:, :S])
        ref_l, ref_g = oracle.rendering_loss(inp, tgt, tab)
        loss, grad = native.rendering_loss(d_in, d_tg, _t(tab, dev))       # with the gradient: table read from memory
        assert_loss_close(loss.item(), ref_l, "S=%d loss" % S)
        assert_grad_close(_np(grad), ref_g, "S=%d loss grad" % S, f64=oracle.rendering_loss(inp, tgt, tab, f64=True)[1],
                          tie_map=oracle.loss_tie_map(inp, tgt, tab), max_ties=48)
        if forward_only_ok:                                                # forward only: the table is staged in LDS
            l2, none = native.rendering_loss(d_in, d_tg, _t(tab, dev), want_grad=False)
            assert none is None
            assert_loss_close(l2.item(), ref_l, "S=%d forward-only loss" % S)
        else:
            with pytest.raises(native.NativeLibraryError, match="too many scenes"):
                native.rendering_loss(d_in, d_tg, _t(tab, dev), want_grad=False)
    # ---- a plane beyond 2^25 pixels: refused by the loss before any launch (nothing is dereferenced)
    need = lib.svbrdf_rendering_loss_workspace_bytes(1, 1, 8192, 8192)
    assert lib.svbrdf_rendering_loss_fwd_bwd(p, p, p, p, ctypes.c_float(0.1), p, p, p, need, 1, 1, 8192, 8192, None) == -2
    assert b"2^25" in lib.svbrdf_last_error()


def test_tied_and_untied_roughness_paths_agree_with_oracle(dev, native, oracle):
    """the kernels take a one-lobe fast path when a whole wave has tied roughness channels:
    all-tied, none-tied and a patch where only some rows are tied (both paths in one launch)"""
    B, H, S = 2, 64, 4
    torch.manual_seed(8)
    from svbrdf_estimation_amd import environment
    table = torch.stack([environment.scene_table(2, 2) for _ in range(B)]).numpy()
    for case in ("tied", "untied", "mixed", "target_untied"):
        inp = synth.make_maps(81, B, H, tiled_roughness=(case != "untied"))
        tgt = synth.make_maps(82, B, H, tiled_roughness=(case in ("tied", "mixed")))
        if case == "mixed":
            inp[:, 7, H // 2:, :] = synth.uniform01(83, (B, H // 2, H))        # lower half: green roughness differs
        ref_l, ref_g = oracle.rendering_loss(inp, tgt, table)
        l, g = native.rendering_loss(_t(inp, dev), _t(tgt, dev), _t(table, dev))
        assert_loss_close(l.item(), ref_l, case)
        assert_grad_close(_np(g), ref_g, case + " grad")
        cot = synth.uniform01(84, (B, S, 3, H, H)) - np.float32(0.5)
        assert_render_strict(_np(native.render_fwd(_t(inp, dev), _t(table, dev))), oracle.render_fwd(inp, table), case)
        assert_grad_close(_np(native.render_bwd(_t(inp, dev), _t(table, dev), _t(cot, dev))),
                          oracle.render_bwd(inp, table, cot), case + " K2")


def test_loss_accumulator_is_left_zeroed_and_reusable(dev, native):
    """the in-kernel finalise (fixed-point atomics + arrival ticket) must re-zero its scratch"""
    inp, tgt = _t(synth.make_maps(91, 3, 40), dev), _t(synth.make_maps(92, 3, 40), dev)
    torch.manual_seed(1)
    from svbrdf_estimation_amd import environment
    table = torch.stack([environment.scene_table(3, 6) for _ in range(3)]).to(dev)
    vals = [native.rendering_loss(inp, tgt, table, want_grad=(i % 2 == 0))[0].item() for i in range(6)]
    assert len(set(vals)) == 1, vals
    for ws in native._workspace_cache.values():
        assert not ws.any().item()


@pytest.mark.parametrize("head", [False, True])
def test_non_finite_maps_give_nan_loss_and_leave_the_scratch_clean(dev, native, head):
    """A NaN / infinite value anywhere in the maps makes the reference's loss non-finite (torch.clamp and log
    propagate it), so isfinite(loss) guards work.  The kernel must do the same -- not cast a NaN partial sum into
    its fixed-point accumulator, not let v_max swallow a NaN normal or roughness -- and the NEXT call on the same
    stream must be bitwise what it was before (scratch left zeroed, arrival counters intact)."""
    from svbrdf_estimation_amd import environment
    B, H = 3, 48
    tgt = synth.make_maps(96, B, H)
    if head:
        clean = (synth.uniform01(97, (B, 9, H, H)) * np.float32(1.8) - np.float32(0.9)).astype(np.float32)
        channels = (0, 1, 3, 5, 7)            # normal xy, diffuse, roughness, specular of the encoded head output
    else:
        clean = synth.make_maps(95, B, H)
        channels = (0, 2, 4, 6, 8, 10)        # normals, diffuse, roughness (tied path broken too), specular
    torch.manual_seed(4)
    table = torch.stack([environment.scene_table(3, 6) for _ in range(B)])
    good_l, good_g = native.rendering_loss(_t(clean, dev), _t(tgt, dev), table, l1_weight=0.1, head=head)
    good_l, good_g = good_l.item(), _np(good_g)
    assert np.isfinite(good_l) and np.isfinite(good_g).all()
    for poison in (np.nan, np.inf, -np.inf):
        for ch in channels:
            bad = clean.copy()
            bad[1, ch, 17, 5] = poison
            l, g = native.rendering_loss(_t(bad, dev), _t(tgt, dev), table, l1_weight=0.1, head=head)
            assert np.isnan(l.item()), "poison %r in input channel %d: loss %r" % (poison, ch, l.item())
            l2, g2 = native.rendering_loss(_t(clean, dev), _t(tgt, dev), table, l1_weight=0.1, head=head)
            assert l2.item() == good_l and np.array_equal(_np(g2), good_g)
    for ch in (1, 3, 7, 11):                  # poisoned TARGET, forward-only kernel and the adjoint kernel
        bad = tgt.copy()
        bad[0, ch, 3, 40] = np.nan
        for want_grad in (False, True):
            l, _ = native.rendering_loss(_t(clean, dev), _t(bad, dev), table, want_grad=want_grad, head=head)
            assert np.isnan(l.item())
    l2, g2 = native.rendering_loss(_t(clean, dev), _t(tgt, dev), table, l1_weight=0.1, head=head)
    assert l2.item() == good_l and np.array_equal(_np(g2), good_g)
    for ws in native._workspace_cache.values():
        assert not ws.any().item()


def test_rendering_loss_module_reproduces_reference_with_same_seed(dev, golden):
    """end to end through RenderingLoss.forward: same torch seed -> same scenes -> same loss"""
    from svbrdf_estimation_amd import losses, renderers
    g = golden("g3_loss_48.npz")
    fn = losses.RenderingLoss(renderers.LocalRenderer())
    x = _t(g["input"], dev).requires_grad_(True)
    torch.manual_seed(int(g["rng_seed"]))
    loss = fn(x, _t(g["target"], dev))
    assert loss.dim() == 0
    (2.0 * loss).backward()
    assert_loss_close(loss.item(), g["loss"], "module loss", rtol=2e-6)
    assert_grad_close(_np(x.grad) / 2.0, g["grad_input"], "module grad (upstream grad 2)")
    # MixedLoss = 0.1 * L1 + rendering (losses.py:54-63): fused path, and target gradient by symmetry
    x2 = _t(g["input"], dev).requires_grad_(True)
    t2 = _t(g["target"], dev).requires_grad_(True)
    torch.manual_seed(int(g["rng_seed"]))
    mfn = losses.MixedLoss(renderers.LocalRenderer())
    mixed = mfn(x2, t2)
    mixed.backward()
    assert_loss_close(mixed.item(), g["mixed_loss"], "mixed", rtol=2e-6)
    assert_grad_close(_np(x2.grad), g["mixed_grad"], "mixed grad")
    from oracle import c_oracle
    _, gt = c_oracle.mixed_loss(g["target"], g["input"], g["scenes"], 0.1)
    _, gt64 = c_oracle.mixed_loss(g["target"], g["input"], g["scenes"], 0.1, f64=True)
    assert_grad_close(_np(t2.grad), gt, "mixed target grad", f64=gt64)
    # l1_weight = 0 degenerates to the rendering loss; the literal (unfused) sum agrees too
    mfn0 = losses.MixedLoss(renderers.LocalRenderer(), l1_weight=0.0)
    torch.manual_seed(int(g["rng_seed"]))
    assert_loss_close(mfn0(_t(g["input"], dev), _t(g["target"], dev)).item(), g["loss"], "mixed w=0", rtol=2e-6)
    torch.manual_seed(int(g["rng_seed"]))
    literal = 0.1 * mfn.l1_loss(_t(g["input"], dev), _t(g["target"], dev)) + mfn.rendering_loss(
        _t(g["input"], dev), _t(g["target"], dev))
    assert_loss_close(literal.item(), mixed.item(), "fused vs literal sum", rtol=2e-6)
    l1 = losses.SVBRDFL1Loss()(_t(g["input"], dev), _t(g["target"], dev))
    assert_loss_close(l1.item(), g["l1_loss"], "l1", rtol=2e-6)


@pytest.mark.parametrize("name", ["g3_loss_48.npz", "g3_loss_7_s5.npz", "g3_loss_20_untied.npz"])
def test_mixed_loss_fused_golden(dev, native, oracle, golden, name):
    """MixedLoss = 0.1 * SVBRDFL1Loss + RenderingLoss in ONE kernel, vs oracle and reference fixture"""
    g = golden(name)
    d_in, d_tg, d_sc = _t(g["input"], dev), _t(g["target"], dev), _t(g["scenes"], dev)
    loss, grad = native.rendering_loss(d_in, d_tg, d_sc, l1_weight=0.1, eps_l1=0.01)
    ref_l, ref_g = oracle.mixed_loss(g["input"], g["target"], g["scenes"], 0.1)
    assert_loss_close(loss.item(), ref_l, name + " vs oracle")
    assert_loss_close(loss.item(), g["mixed_loss"], name + " vs reference", rtol=2e-6)
    _, g64 = oracle.mixed_loss(g["input"], g["target"], g["scenes"], 0.1, f64=True)
    assert_grad_close(_np(grad), ref_g, name + " grad vs oracle", f64=g64)
    assert_grad_close(_np(grad), g["mixed_grad"], name + " grad vs reference", f64=g64)
    # the L1 part alone (weight 1, rendering loss subtracted) against the reference's SVBRDFL1Loss
    l_w1, _ = native.rendering_loss(d_in, d_tg, d_sc, l1_weight=1.0, want_grad=False)
    l_w0, _ = native.rendering_loss(d_in, d_tg, d_sc, want_grad=False)
    assert abs((l_w1.item() - l_w0.item()) - float(g["l1_loss"])) <= 3e-6 * float(g["l1_loss"])
    # identical maps: exactly zero, gradient exactly zero
    l0, g0 = native.rendering_loss(d_in, d_in.clone(), d_sc, l1_weight=0.1)
    assert l0.item() == 0.0 and not g0.any().item()


def test_rendering_loss_custom_scene_counts_and_target_grad(dev, oracle, golden):
    from svbrdf_estimation_amd import losses, renderers
    g = golden("g3_loss_7_s5.npz")
    fn = losses.RenderingLoss(renderers.LocalRenderer())
    fn.random_configuration_count, fn.specular_configuration_count = int(g["n_random"]), int(g["n_specular"])
    x = _t(g["input"], dev).requires_grad_(True)
    t = _t(g["target"], dev).requires_grad_(True)
    torch.manual_seed(int(g["rng_seed"]))
    loss = fn(x, t)
    loss.backward()
    assert_loss_close(loss.item(), g["loss"], "s5 loss", rtol=2e-6)
    assert_grad_close(_np(x.grad), g["grad_input"], "s5 grad")
    _, gt = oracle.rendering_loss(g["target"], g["input"], g["scenes"])
    assert_grad_close(_np(t.grad), gt, "target grad (roles swapped)")


def test_config2_full_size_properties_and_all_8_items_vs_oracle(dev, native, oracle):
    """BASELINE config 2 at its size (B=8, 256x256, S=9): size-independent properties, and every pixel of ALL 8 items
    against the oracle (until round 4: items 0 and 7 only)."""
    from svbrdf_estimation_amd import losses, renderers
    B, H = 8, 256
    inp, tgt = synth.make_maps(61, B, H), synth.make_maps(62, B, H)
    fn = losses.RenderingLoss(renderers.LocalRenderer())
    torch.manual_seed(2024)
    table = fn.sample_scene_table(B).numpy()
    d_in, d_tg, d_sc = _t(inp, dev), _t(tgt, dev), _t(table, dev)
    loss, grad = native.rendering_loss(d_in, d_tg, d_sc)
    # (1) identical input and target: loss exactly 0, gradient exactly 0 (sign(0) = 0)
    l0, g0 = native.rendering_loss(d_in, d_in.clone(), d_sc)
    assert l0.item() == 0.0 and not g0.any().item()
    # (2) symmetry of |log a - log b| in the two arguments
    ls, _ = native.rendering_loss(d_tg, d_in, d_sc)
    assert_loss_close(ls.item(), loss.item(), "symmetry", rtol=1e-6)
    # (3) bitwise run-to-run determinism
    l2, g2 = native.rendering_loss(d_in, d_tg, d_sc)
    assert l2.item() == loss.item() and torch.equal(g2, grad)
    # (4) the loss of the batch is the mean of the per-item losses (what data-parallel sharding relies on)
    per_item = [native.rendering_loss(d_in[b:b + 1], d_tg[b:b + 1], d_sc[b:b + 1])[0].item() for b in range(B)]
    assert_loss_close(np.mean(per_item), loss.item(), "mean of shards", rtol=1e-6)
    # (5) the fused kernel agrees with the separate K1 renderings pushed through torch's log/L1
    ri, rt = native.render_fwd(d_in, d_sc), native.render_fwd(d_tg, d_sc)
    l_sep = (torch.log(ri.double() + 0.1) - torch.log(rt.double() + 0.1)).abs().mean()
    assert_loss_close(loss.item(), l_sep.item(), "fused vs K1+torch", rtol=2e-6)
    # (6) and K2 applied to the L1 cotangent reproduces the fused gradient
    cot = (torch.sign(torch.log(ri + 0.1) - torch.log(rt + 0.1)) / (ri + 0.1) / ri.numel())
    assert_grad_close(_np(native.render_bwd(d_in, d_sc, cot)), _np(grad), "fused grad vs K2")
    # (7) the oracle, item by item (per-item loss) and on the whole batch (gradient with the fp64 widening and the tie map)
    oracle.set_threads(min(32, os.cpu_count() or 1))
    for b in range(B):
        lo, _ = oracle.rendering_loss(inp[b:b + 1], tgt[b:b + 1], table[b:b + 1])
        assert_loss_close(per_item[b], lo, "config-2 item %d vs oracle" % b)
    ref_l, ref_g = oracle.rendering_loss(inp, tgt, table)
    _, g64 = oracle.rendering_loss(inp, tgt, table, f64=True)
    assert_loss_close(loss.item(), ref_l, "config-2 batch loss vs oracle")
    # every gradient element of all 8 items; expected ties ~ 2e-6 per (pixel, scene, channel) term, as at config 5
    n_terms = B * H * H * table.shape[1] * 3
    assert_grad_close(_np(grad), ref_g, "config-2 gradient, all 8 items", f64=g64, tie_map=oracle.loss_tie_map(inp, tgt, table),
                      max_ties=max(8, int(2e-6 * n_terms)), max_widened=max(8, int(2e-6 * grad.numel())))


def test_config5_per_gpu_shape_512_32_scenes_batch_8_vs_oracle(dev, native, oracle):
    """BASELINE config 5 at its per-GPU size: batch 8 of 512x512 patches, 32 scenes (11 random + 21 specular -- the
    21-element normal_ draws take torch's vectorised path), mixed loss; every pixel of all 8 items against the oracle
    (until round 4: B = 2)"""
    from svbrdf_estimation_amd import losses, renderers
    B, H = 8, 512
    inp, tgt = synth.make_maps(95, B, H), synth.make_maps(96, B, H)
    fn = losses.MixedLoss(renderers.LocalRenderer())
    fn.rendering_loss.random_configuration_count, fn.rendering_loss.specular_configuration_count = 11, 21
    torch.manual_seed(55)
    table = fn.rendering_loss.sample_scene_table(B).numpy()
    assert table.shape == (B, 32, 9)
    x = _t(inp, dev).requires_grad_(True)
    torch.manual_seed(55)
    loss = fn(x, _t(tgt, dev))
    loss.backward()
    oracle.set_threads(min(32, os.cpu_count() or 1))
    ref_l, ref_g = oracle.mixed_loss(inp, tgt, table, 0.1)
    _, g64 = oracle.mixed_loss(inp, tgt, table, 0.1, f64=True)
    assert_loss_close(loss.item(), ref_l, "config-5 mixed loss")
    # expected ties ~ 2e-6 per (pixel, scene, channel) term; measured 0.6e-6 (31 of 524288 pixels at 512x512, 32 scenes); the cap allows 2e-6
    assert_grad_close(_np(x.grad), ref_g, "config-5 gradient, all 8 items", f64=g64, tie_map=oracle.loss_tie_map(inp, tgt, table),
                      max_ties=max(8, int(2e-6 * inp.shape[0] * inp.shape[2] * inp.shape[3] * table.shape[1] * 3)),
                      max_widened=max(8, int(2e-6 * inp.size)))


# ---------------------------------------------------------------- plugin interface (renderers.py:67)

def test_large_patch_1024(dev, native, oracle):
    """a 1024x1024 patch (4x the pixels of config 5, 48 MB per map set): K1, K2 and the fused loss against
    the oracle at every pixel -- row/column indexing and plane strides beyond the sizes the reference trains at"""
    from svbrdf_estimation_amd import environment
    B, S, H = 1, 2, 1024
    maps, tgt = synth.make_maps(201, B, H, tiled_roughness=False), synth.make_maps(202, B, H)
    torch.manual_seed(3)
    table = torch.stack([environment.scene_table(1, 1) for _ in range(B)]).numpy()
    cot = synth.uniform01(9, (B, S, 3, H, H)) - np.float32(0.5)
    oracle.set_threads(min(32, os.cpu_count() or 1))
    assert_render_strict(_np(native.render_fwd(_t(maps, dev), _t(table, dev))), oracle.render_fwd(maps, table), "1024 fwd")
    assert_grad_close(_np(native.render_bwd(_t(maps, dev), _t(table, dev), _t(cot, dev))),
                      oracle.render_bwd(maps, table, cot), "1024 bwd")
    loss, grad = native.rendering_loss(_t(maps, dev), _t(tgt, dev), torch.from_numpy(table))
    ref_l, ref_g = oracle.rendering_loss(maps, tgt, table)
    _, g64 = oracle.rendering_loss(maps, tgt, table, f64=True)
    assert_loss_close(loss.item(), ref_l, "1024 loss")
    assert_grad_close(_np(grad), ref_g, "1024 loss grad", f64=g64, tie_map=oracle.loss_tie_map(maps, tgt, table),
                      max_ties=max(8, int(2e-6 * maps.shape[0] * maps.shape[2] * maps.shape[3] * table.shape[1] * 3)))


def test_local_renderer_interface(dev, oracle, golden):
    from svbrdf_estimation_amd import environment as env
    from svbrdf_estimation_amd import renderers
    g = golden("g4_batched_one_scene.npz")
    R = renderers.LocalRenderer()
    sc = env.Scene(env.Camera(torch.tensor(g["scene"][0:3])),
                   env.Light(torch.tensor(g["scene"][3:6]), torch.tensor(g["scene"][6:9])))
    x = _t(g["maps"], dev).requires_grad_(True)
    out = R.render(sc, x)                       # 4-D input, one scene for the whole batch
    assert tuple(out.shape) == (3, 3, 16, 16)
    out.backward(_t(g["cot"], dev))
    scn = np.repeat(g["scene"][None, None], 3, 0)
    assert_render_vs_reference(_np(out), g["out"], oracle.render_fwd(g["maps"], scn, f64=True)[:, 0], "batched")
    assert_grad_close(_np(x.grad), g["grad"], "batched grad")
    out3 = R.render(sc, _t(g["maps"][0], dev))  # 3-D input -> [1,3,H,W]
    assert tuple(out3.shape) == tuple(g["out3_shape"])
    assert torch.equal(out3[0], out[0].detach())
    # list / ndarray positions (losses.py callers) and a non-contiguous view
    sc2 = env.Scene(env.Camera(list(g["scene"][0:3])), env.Light(np.asarray(g["scene"][3:6]), list(g["scene"][6:9])))
    wide = torch.zeros(3, 12, 16, 32, device=dev)
    wide[..., ::2] = _t(g["maps"], dev)
    assert torch.equal(R.render(sc2, wide[..., ::2]), out.detach())
    # a HOST tensor (the reference dataloader's call, dataset.py:206-212) is rendered on the GPU and comes back a host tensor;
    # what the engine cannot do for it still fails loudly: a gradient, a dtype it has no kernel for
    host = torch.from_numpy(g["maps"])
    assert R.render(sc, host).device.type == "cpu" and torch.equal(R.render(sc, host), out.detach().cpu())
    with pytest.raises(Exception):
        R.render(sc, host.clone().requires_grad_(True))
    with pytest.raises(TypeError):
        R.render(sc, torch.zeros(12, 8, 8, dtype=torch.float16))
    with pytest.raises(ValueError):
        R.render(sc, torch.zeros(12, 8, 4, device=dev))         # H != W
    with pytest.raises(TypeError):
        R.render(sc, torch.zeros(12, 8, 8, device=dev, dtype=torch.float16))    # float32 and float64 only (test_gpu_float64.py)
    assert R.render(sc, torch.zeros(12, 8, 8, device=dev, dtype=torch.float64)).dtype == torch.float64
    many = R.render_many(torch.from_numpy(np.repeat(g["scene"][None], 2, 0)), _t(g["maps"], dev))
    assert tuple(many.shape) == (3, 2, 3, 16, 16) and torch.equal(many[:, 1], out.detach())


def test_render_native_and_ctypes_host_paths_are_bitwise_identical(dev, golden):
    """LocalRenderer.render through csrc/host_ext.cpp (C++ autograd node) and through the Python/ctypes
    autograd.Function: same kernels, same bits, forward and backward; non-contiguous maps, a non-leaf input, repeated
    backward under retain_graph, a host tensor staged through K1, and the loud failure on fp16 tensors in both"""
    from svbrdf_estimation_amd import _hostext, environment as env, renderers
    assert _hostext.module() is not None
    g = golden("g4_batched_one_scene.npz")
    R = renderers.LocalRenderer()
    sc = env.Scene(env.Camera(list(g["scene"][0:3])), env.Light(list(g["scene"][3:6]), list(g["scene"][6:9])))
    cot = _t(g["cot"], dev)
    res = {}
    for name, on in (("native", True), ("ctypes", False)):
        _hostext.set_enabled(on)
        try:
            x = _t(g["maps"], dev).requires_grad_(True)
            out = R.render(sc, x)
            node = out.grad_fn.next_functions[0][0]                    # below the final .view of render()
            assert ("SvbrdfRenderBackward" in type(node).__name__ or "SvbrdfRenderBackward" in node.name()) == on, node.name()
            out.backward(cot, retain_graph=True)
            first = x.grad.clone()
            out.backward(cot)                                           # second backward: accumulates
            w = torch.ones(1, device=dev, requires_grad=True)
            (gw,) = torch.autograd.grad(R.render(sc, _t(g["maps"], dev) * w).sum(), w)
            wide = torch.zeros(3, 12, 16, 32, device=dev)
            wide[..., ::2] = _t(g["maps"], dev)
            res[name] = (out.detach(), first, x.grad.clone(), gw, R.render(sc, wide[..., ::2]))
            assert torch.equal(R.render(sc, torch.from_numpy(g["maps"])), out.detach().cpu())       # host tensor: staged, K1
            with pytest.raises(Exception):
                R.render(sc, torch.zeros(12, 8, 8, dtype=torch.float16))
            with pytest.raises(Exception):
                R.render(sc, torch.zeros(12, 8, 8, device=dev, dtype=torch.float16))
        finally:
            _hostext.set_enabled(True)
    for a, b in zip(res["native"], res["ctypes"]):
        assert torch.equal(a, b)
    assert torch.equal(res["native"][2], res["native"][1] + res["native"][1])
    assert torch.equal(res["native"][4], res["native"][0])


def test_plugin_path_with_foreign_renderer(dev):
    """RenderingLoss keeps the duck-typed plugin protocol for any other renderer object"""
    from svbrdf_estimation_amd import losses

    class Flat:
        def render(self, scene, svbrdf):
            return svbrdf[3:6].unsqueeze(0) * float(scene.light.color[0])

    x = torch.rand(2, 12, 8, 8, device=dev, requires_grad=True)
    t = torch.rand(2, 12, 8, 8, device=dev)
    torch.manual_seed(0)
    loss = losses.RenderingLoss(Flat())(x, t)
    loss.backward()
    assert loss.item() > 0 and x.grad[:, 3:6].abs().sum().item() > 0 and not x.grad[:, 0:3].any().item()


def test_optimisation_through_the_loss_decreases(dev):
    """the notebooks' experiment (website.ipynb:300-337): optimise raw maps through the loss"""
    from svbrdf_estimation_amd import losses, renderers
    tgt = _t(synth.make_maps(71, 1, 32), dev)
    x = _t(synth.make_maps(72, 1, 32), dev).requires_grad_(True)
    fn = losses.RenderingLoss(renderers.LocalRenderer())
    opt = torch.optim.Adam([x], lr=0.02)
    torch.manual_seed(3)
    table = fn.sample_scene_table(1).to(dev)
    first = last = None
    for _ in range(60):
        opt.zero_grad()
        loss = losses._FusedRenderingLoss.apply(x, tgt, table, 0.1)
        loss.backward()
        opt.step()
        with torch.no_grad():                       # keep the maps in their physical range
            x[:, 3:].clamp_(0.01, 1.0)
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < 0.6 * first


# ---------------------------------------------------------------- row f3: input-photo synthesis on the GPU

def test_render_inputs_matches_reference_dataloader(dev, oracle, golden):
    """dataset.py:162-221 on the GPU: same scenes, same (CPU-drawn) noise field, same photos, and the
    CPU generator ends in the same state as after the reference's call"""
    from svbrdf_estimation_amd import synthesis
    g = golden("g10_render_inputs.npz")
    for aug in (0, 1):
        for n in (1, 4):
            k = "aug%d_n%d" % (aug, n)
            torch.manual_seed(int(g[k + "__seed"]))
            out = synthesis.render_inputs(_t(g[k + "__maps"], dev), n, use_augmentation=bool(aug), noise="cpu")
            assert tuple(out.shape) == (n, 3, 32, 32)
            assert np.array_equal(torch.get_rng_state().numpy()[:64], g[k + "__rng_after"]), k
            ref = g[k + "__out"]
            err = np.abs(_np(out) - ref)
            # photos are clamped to [0,1]; a highlight pixel carries the reference's own sqrt noise, so:
            # at most MAX_WIDENED_RENDER pixels of a case (counted, printed) outside 1e-5 rel + 2e-6, every pixel
            # within 3e-5 absolute
            from tolerances import MAX_WIDENED_RENDER, _record
            outside = int((err > 1e-5 * np.abs(ref) + 2e-6).sum())
            _record("render_inputs " + k, "outside 1e-5 rel + 2e-6", outside, err.size, min(MAX_WIDENED_RENDER, 8 * n))
            assert outside <= min(MAX_WIDENED_RENDER, 8 * n), (k, outside, err.max())
            assert err.max() <= 3e-5, (k, err.max())
            assert out.min().item() >= 0.0 and out.max().item() <= 1.0
    # batched: B samples x count photos in one launch; device noise is statistically right
    maps = _t(synth.make_maps(510, 3, 32), dev)
    torch.manual_seed(1)
    clean = synthesis.render_inputs(maps, 2, noise=None)
    torch.manual_seed(1)
    noisy = synthesis.render_inputs(maps, 2, noise="device")
    assert tuple(noisy.shape) == (3, 2, 3, 32, 32)
    inner = (clean > 0.05) & (clean < 0.95)
    resid = (noisy - clean)[inner]
    assert abs(resid.mean().item()) < 2e-3 and 1e-3 < resid.std().item() < 3e-2


def test_render_inputs_fused_noise_and_clamp_epilogue(dev):
    """SURVEY section 8 row f3 "K1 (+ Gaussian noise, clamp [0,1])" (dataset.py:215-217) as ONE launch,
    svbrdf_render_inputs*: without levels it is bitwise clamp(K1); with levels the photo is
    clamp(K1 + sigma_image * n) where n is the counter-based field the header defines -- checked element by element against
    the numpy restatement of Philox4x32-10 + Box-Muller (tests/philox_ref.py, pinned by the published vectors in the CPU
    suite) for every vector width, for host and device tables, and statistically per image."""
    import philox_ref
    from svbrdf_estimation_amd import _native, synthesis
    seed, offset = 0x1234ABCD5678, 0x100000040
    for H, B, S in ((32, 3, 2), (30, 2, 3), (31, 2, 2), (64, 2, 5)):            # W % 4 == 0 / % 2 == 0 / odd: VEC 4, 2, 1
        maps = _t(synth.make_maps(900 + H, B, H), dev)
        torch.manual_seed(H)
        table = torch.stack([synthesis.input_scene_table(S, True) for _ in range(B)])
        levels = (synthesis.noise_levels(B * S) * 4.0).view(B, S)               # 0.01 .. 0.05: well above fp32 resolution
        plain = _native.render_fwd(maps, table)
        # no levels: clamp only, bit for bit, host table and device table
        assert torch.equal(_native.render_inputs(maps, table), plain.clamp(0.0, 1.0))
        assert torch.equal(_native.render_inputs(maps, table.to(dev)), plain.clamp(0.0, 1.0))
        noisy = _native.render_inputs(maps, table, levels, seed, offset)
        assert torch.equal(noisy, _native.render_inputs(maps, table.to(dev), levels.to(dev), seed, offset))
        assert torch.equal(noisy, _native.render_inputs(maps, table, levels, seed, offset))      # a pure function of its inputs
        assert not torch.equal(noisy, _native.render_inputs(maps, table, levels, seed, offset + 4))
        assert not torch.equal(noisy, _native.render_inputs(maps, table, levels, seed + 1, offset))
        field = philox_ref.normal_field(seed, offset, plain.numel()).reshape(plain.shape)
        want = np.clip(_np(plain).astype(np.float64) + levels.numpy().astype(np.float64)[:, :, None, None, None] * field, 0.0, 1.0)
        err = np.abs(_np(noisy) - want)
        # v_log / v_sqrt / v_sin / v_cos are ~1 ulp of their results; the field enters scaled by sigma <= 0.05
        assert err.max() <= 2e-7 + 3e-6 * float(levels.max()), (H, err.max())
        assert noisy.min().item() >= 0.0 and noisy.max().item() <= 1.0
        # statistics per image, on pixels no clamp touched: residual / sigma ~ N(0,1).  (4 sigma per image: this test makes
        # ~150 such comparisons, which is 3 sigma family-wise; the element-by-element check above is the sharp one)
        for b in range(B):
            for s_ in range(S):
                clean = _np(plain[b, s_])
                inner = (clean > 0.3) & (clean < 0.7)
                if inner.sum() < 200:
                    continue
                z = (_np(noisy[b, s_])[inner] - clean[inner]) / float(levels[b, s_])
                n = z.size
                assert abs(z.mean()) < 4.0 / np.sqrt(n) and abs(z.var() - 1.0) < 4.0 * np.sqrt(2.0 / n), (H, b, s_, z.mean(), z.var())
    # BASELINE shape (B = 8, 256x256, 5 photos per sample = configs[3]'s view count): the field does not depend on the vector
    # width the launch picked (SVBRDF_K1_VEC forces the narrower kernels: same bits), clamp-only is bitwise clamp(K1), every
    # image's residual is standard normal where no clamp touched it
    maps = _t(synth.make_maps(970, 8, 256), dev)
    torch.manual_seed(256)
    table = torch.stack([synthesis.input_scene_table(5, True) for _ in range(8)])
    levels = synthesis.noise_levels(40).view(8, 5)
    plain = _native.render_fwd(maps, table)
    assert torch.equal(_native.render_inputs(maps, table), plain.clamp(0.0, 1.0))
    wide = _native.render_inputs(maps, table, levels, 77, 12)
    try:
        for vec in ("2", "1"):
            os.environ["SVBRDF_K1_VEC"] = vec
            assert torch.equal(_native.render_inputs(maps, table, levels, 77, 12), wide), vec
    finally:
        os.environ.pop("SVBRDF_K1_VEC", None)
    clean, got = _np(plain), _np(wide)
    for b in range(8):
        for s_ in range(5):
            inner = (clean[b, s_] > 0.2) & (clean[b, s_] < 0.8)
            if inner.sum() < 2000:
                continue
            z = (got[b, s_][inner] - clean[b, s_][inner]) / float(levels[b, s_])
            assert abs(z.mean()) < 4.0 / np.sqrt(z.size) and abs(z.var() - 1.0) < 4.0 * np.sqrt(2.0 / z.size), (b, s_)
            assert abs((z ** 4).mean() - 3.0) < 0.25
    # NaN maps stay NaN through the clamp (torch.clamp's behaviour), like clamp(K1)
    bad = _t(synth.make_maps(950, 1, 16), dev)
    bad[0, 4, 3, 5] = float("nan")
    tab = torch.stack([synthesis.input_scene_table(1, False)])
    a, b = _native.render_inputs(bad, tab), _native.render_fwd(bad, tab).clamp(0.0, 1.0)
    assert torch.isnan(a).sum().item() == torch.isnan(b).sum().item() > 0 and torch.equal(torch.nan_to_num(a, 7.0), torch.nan_to_num(b, 7.0))
    # through synthesis.render_inputs: ONE kernel launch per call; keyed by torch's device generator
    maps = _t(synth.make_maps(960, 4, 32), dev)
    for noise in ("device", None):
        before = _native.launch_count()
        synthesis.render_inputs(maps, 3, noise=noise)
        assert _native.launch_count() - before == 1, noise
    outs = []
    for _ in range(2):
        torch.manual_seed(3)
        torch.cuda.manual_seed(77)
        outs.append((synthesis.render_inputs(maps, 2), synthesis.render_inputs(maps, 2)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    torch.manual_seed(3)
    torch.cuda.manual_seed(77)
    first = synthesis.render_inputs(maps, 2)
    torch.manual_seed(3)                                 # same scenes and levels, the device generator has moved on
    assert not torch.equal(first, synthesis.render_inputs(maps, 2))
    # more photos than the argument block holds: tables are uploaded, same field
    big = _t(synth.make_maps(961, 40, 16), dev)
    torch.manual_seed(9)
    table = torch.stack([synthesis.input_scene_table(8, True) for _ in range(40)])       # 320 rows > 288
    levels = synthesis.noise_levels(320).view(40, 8)
    assert 40 * 8 > _native.host_scenes_max_rows()
    via_upload = _native.render_inputs(big, table, levels, 5, 8)
    assert torch.equal(via_upload, _native.render_inputs(big, table.to(dev), levels.to(dev), 5, 8))
    lib = _native._load()
    out = torch.empty_like(via_upload)
    rc = lib.svbrdf_render_inputs_host_scenes(big.data_ptr(), table.data_ptr(), levels.data_ptr(), 5, 8,
                                               _native.xrow(dev, 16).data_ptr(), out.data_ptr(), 40, 8, 16, 16, None)
    assert rc == -2 and b"SVBRDF_HOST_SCENES_MAX_ROWS" in lib.svbrdf_last_error()


def test_render_inputs_noise_field_seeded_sweep_of_shapes_seeds_and_offsets(dev):
    """the noise epilogue over a seeded sweep of odd shapes (W from 1 to 12: every vector width, planes that are not a
    multiple of four elements), batch and photo counts, full-range 64-bit seeds and offsets: element by element the numpy
    restatement's field, for host and device tables alike"""
    import philox_ref
    from svbrdf_estimation_amd import _native, synthesis
    rng = np.random.default_rng(2026)
    for case in range(40):
        W = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12]))
        B, S = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        seed = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        offset = int(rng.integers(0, 2 ** 62)) * 4
        maps = _t(synth.make_maps(3000 + case, B, W), dev)
        torch.manual_seed(case)
        table = torch.stack([synthesis.input_scene_table(S, bool(case % 2)) for _ in range(B)])
        levels = (synthesis.noise_levels(B * S) * 6.0).view(B, S)
        plain = _native.render_fwd(maps, table)
        noisy = _native.render_inputs(maps, table, levels, seed, offset)
        assert torch.equal(noisy, _native.render_inputs(maps, table.to(dev), levels.to(dev), seed, offset)), case
        field = philox_ref.normal_field(seed, offset, plain.numel()).reshape(plain.shape)
        want = np.clip(_np(plain).astype(np.float64) + levels.numpy().astype(np.float64)[:, :, None, None, None] * field, 0.0, 1.0)
        err = np.abs(_np(noisy) - want)
        assert err.max() <= 2e-7 + 3e-6 * float(levels.max()), (case, W, B, S, err.max())


def test_render_of_a_host_tensor_is_the_device_call_and_serves_the_reference_dataloader(dev, golden):
    """the reference's SECOND caller of ``render`` is its dataloader, in the main process, with CPU
    tensors -- ``renderer.render(scene, svbrdf.unsqueeze(0))`` with tensor-valued positions and colour, CPU noise added to
    the CPU result, ``torch.cat`` of CPU photos (dataset.py:94-98, :206-221).  The call is served by K1 through a pinned
    round trip: the result is a CPU tensor, bit for bit the device call's, and the reference's loop, restated here call by
    call around it, reproduces the photos AND the generator state of the reference's own run (g10)."""
    from svbrdf_estimation_amd import NativeLibraryError, environment as env, renderers, synthesis
    from tolerances import MAX_WIDENED_RENDER, _record
    g = golden("g10_render_inputs.npz")
    renderer = renderers.LocalRenderer()
    for aug in (0, 1):
        for count in (1, 4):
            k = "aug%d_n%d" % (aug, count)
            svbrdf = torch.from_numpy(g[k + "__maps"])                       # [12,32,32] on the HOST, like dataset.py:85
            torch.manual_seed(int(g[k + "__seed"]))
            table = synthesis.input_scene_table(count, bool(aug))            # dataset.py:172-204, the reference's draws
            view_poses, light_poses, light_colors = table[:, 0:3], table[:, 3:6], table[:, 6:9]
            renderings = []
            for i in range(count):                                           # dataset.py:206-219, call for call
                scene = env.Scene(env.Camera(view_poses[i]), env.Light(light_poses[i], light_colors[i]))
                rendering = renderer.render(scene, svbrdf.unsqueeze(0))
                assert rendering.device.type == "cpu" and tuple(rendering.shape) == (1, 3, 32, 32)
                assert not rendering.is_pinned() and not rendering.requires_grad
                on_device = renderer.render(scene, svbrdf.unsqueeze(0).to(dev))
                assert torch.equal(rendering, on_device.cpu()), (k, i)
                std = torch.exp(torch.Tensor(1).normal_(mean=np.log(0.005), std=0.3)).numpy()[0]
                noise = torch.zeros_like(rendering).normal_(mean=0.0, std=std)
                renderings.append(torch.clamp(rendering + noise, min=0.0, max=1.0))
            out = torch.cat(renderings, dim=0)
            assert np.array_equal(torch.get_rng_state().numpy()[:64], g[k + "__rng_after"]), k
            ref = g[k + "__out"]
            err = np.abs(out.numpy() - ref)
            outside = int((err > 1e-5 * np.abs(ref) + 2e-6).sum())           # same bound as the device synthesis above
            _record("host-tensor render, dataloader loop " + k, "outside 1e-5 rel + 2e-6", outside, err.size,
                    min(MAX_WIDENED_RENDER, 8 * count))
            assert outside <= min(MAX_WIDENED_RENDER, 8 * count), (k, outside, err.max())
            assert err.max() <= 3e-5, (k, err.max())
    # shapes and views the staging must gather: 3-D input, a batch with the one scene, a strided view, a second call
    # reusing the slot with other data, 256x256
    scene = env.Scene(env.Camera([0.1, -0.2, 2.0]), env.Light(np.array([0.4, 0.3, 1.5]), torch.tensor([30.0, 28.0, 31.0])))
    for shape in ((12, 16, 16), (3, 12, 16, 16), (1, 12, 256, 256)):
        for seed in (1, 2):
            host = torch.from_numpy(synth.make_maps(700 + seed, shape[0] if len(shape) == 4 else 1, shape[-1]))
            host = host[0] if len(shape) == 3 else host
            got = renderer.render(scene, host)
            assert got.device.type == "cpu" and torch.equal(got, renderer.render(scene, host.to(dev)).cpu())
    wide = torch.from_numpy(synth.make_maps(703, 2, 24))
    strided = wide.transpose(-1, -2)                                         # a non-contiguous view: gathered by the host copy
    assert not strided.is_contiguous()
    assert torch.equal(renderer.render(scene, strided), renderer.render(scene, strided.to(dev)).cpu())
    dbl = wide.double()
    got = renderer.render(scene, dbl)
    assert got.dtype == torch.float64 and got.device.type == "cpu"
    assert torch.equal(got, renderer.render(scene, dbl.to(dev)).cpu())
    # forward only: a host tensor that wants a gradient is refused, never computed somewhere else
    with pytest.raises(NativeLibraryError):
        renderer.render(scene, wide.clone().requires_grad_(True))
    with torch.no_grad():                                                    # ... unless nobody records a graph
        assert torch.equal(renderer.render(scene, wide.clone().requires_grad_(True)), renderer.render(scene, wide))


# ---------------------------------------------------------------- native host path vs ctypes path

def test_host_extension_and_ctypes_paths_are_bitwise_identical(dev, golden):
    """csrc/host_ext.cpp (C++ sampler + autograd node) and the Python/ctypes path draw the same
    scenes, launch the same kernels and leave the CPU generator in the same state"""
    from svbrdf_estimation_amd import _hostext, losses, renderers
    assert _hostext.module() is not None, "host extension not built (python -c 'import __graft_entry__ as g; g.build()')"
    g = golden("g3_loss_48.npz")
    res = {}
    try:
        for name, enabled in (("ext", True), ("ctypes", False)):
            _hostext.set_enabled(enabled)
            assert (_hostext.module() is not None) == enabled
            out = []
            for fn in (losses.RenderingLoss(renderers.LocalRenderer()), losses.MixedLoss(renderers.LocalRenderer())):
                x = _t(g["input"], dev).requires_grad_(True)
                t = _t(g["target"], dev).requires_grad_(True)
                torch.manual_seed(int(g["rng_seed"]))
                loss = fn(x, t)
                (3.0 * loss).backward()
                out += [loss.detach().clone(), x.grad.clone(), t.grad.clone(), torch.get_rng_state().clone()]
            res[name] = out
    finally:
        _hostext.set_enabled(True)
    for a, b in zip(res["ext"], res["ctypes"]):
        assert torch.equal(a, b)
    assert_loss_close(res["ext"][0].item(), g["loss"], "ext path loss", rtol=2e-6)
    assert_grad_close(_np(res["ext"][1]) / 3.0, g["grad_input"], "ext path grad")
    # the C++ sampler alone: identical to the per-item reference-order draws on THIS machine, and
    # equal to the fixture up to the CPU math library (torch's CPU sqrt/cos go through MKL, whose
    # code path -- hence last bit -- depends on the host CPU; the fixture comes from the build box)
    from svbrdf_estimation_amd import environment
    g5 = golden("g5_scene_sampler.npz")
    torch.manual_seed(99)
    per_item = torch.stack([environment.scene_table(3, 6) for _ in range(2)])
    torch.manual_seed(99)
    native_tab = _hostext.module().sample_scene_table(2, 3, 6)
    assert torch.equal(native_tab, per_item)
    np.testing.assert_allclose(native_tab.numpy(), g5["seed_99_two_items"], rtol=3e-7, atol=1e-7)


def test_retain_graph_allows_a_second_backward_like_plain_autograd(dev, golden):
    """losses.py:29-52 is plain autograd in the reference, so ``loss.backward(retain_graph=True)`` followed by another
    backward works there.  Here the kernel's gradient buffer is normally MOVED to the caller and scaled in place; with
    retain_graph=True it stays with the graph and each backward receives a scaled copy (the ctypes fallback cloned on
    every backward until round 4: an extra 25 MB pass per step).  Both host paths; a backward after a non-retaining one
    fails loudly in both, as with any freed graph."""
    from svbrdf_estimation_amd import _hostext, environment, losses, renderers
    g = golden("g3_loss_48.npz")
    d_tg = _t(g["target"], dev)
    fn = losses.RenderingLoss(renderers.LocalRenderer())

    def via_module(x):
        torch.manual_seed(3)
        return fn(x, d_tg)

    def via_ctypes(x):
        torch.manual_seed(3)
        return losses._FusedRenderingLoss.apply(x, d_tg, fn.sample_scene_table(x.shape[0]), 0.1)

    assert _hostext.module() is not None
    for make in (via_module, via_ctypes):
        x = _t(g["input"], dev).requires_grad_(True)
        make(x).backward()
        once = x.grad.clone()
        x = _t(g["input"], dev).requires_grad_(True)
        loss = make(x)
        loss.backward(retain_graph=True)
        assert torch.equal(x.grad, once)
        loss.backward(torch.tensor(0.5, device=dev), retain_graph=True)        # a scaled copy, the buffer is untouched
        assert torch.equal(x.grad, once + 0.5 * once)
        loss.backward()                                                          # last one: may consume the buffer
        assert torch.equal(x.grad, (once + 0.5 * once) + once)
        with pytest.raises(RuntimeError):      # both nodes have given their buffer away: a freed graph, and it says so
            loss.backward()
        w = torch.full((1,), 1.0, device=dev, requires_grad=True)               # non-leaf input (the training case)
        loss = make(_t(g["input"], dev) * w)
        (g1,) = torch.autograd.grad(loss, w, retain_graph=True)
        (g2,) = torch.autograd.grad(loss, w)
        assert torch.equal(g1, g2) and torch.isfinite(g1).all()


def test_host_scene_table_rides_in_the_kernel_arguments(dev, native, golden):
    """svbrdf_*_host_scenes: a table of <= 288 rows in HOST memory is passed by value with the launch.
    Same kernels body, so bitwise the same loss and gradient as the device-table entry points; the
    host buffer is consumed before the call returns; larger tables are refused by the C ABI and
    uploaded transparently by the Python wrapper."""
    lib = native._load()
    cap = native.host_scenes_max_rows()
    assert cap == 288
    from svbrdf_estimation_amd import environment
    torch.manual_seed(5)
    for B, S, H in ((3, 5, 20), (8, 12, 8), (1, 288, 9), (16, 18, 8)):          # 15, 96, exactly 288 rows twice
        table = torch.stack([environment.scene_table(S // 2, S - S // 2) for _ in range(B)])
        assert B * S <= cap and not table.is_cuda
        tgt = _t(synth.make_maps(3 * H, B, H), dev)
        maps = _t(synth.make_maps(3 * H + 1, B, H, tiled_roughness=False), dev)
        enc = _t(synth.uniform01(H, (B, 9, H, H)) * 2 - 1, dev)
        for head in (False, True):
            for l1w in (0.0, 0.1):
                for want_grad in (True, False):
                    x = enc if head else maps
                    l_dev, g_dev = native.rendering_loss(x, tgt, table.to(dev), want_grad=want_grad, l1_weight=l1w, head=head)
                    l_host, g_host = native.rendering_loss(x, tgt, table, want_grad=want_grad, l1_weight=l1w, head=head)
                    assert torch.equal(l_dev, l_host), (B, S, head, l1w, want_grad)
                    if want_grad:
                        assert torch.equal(g_dev, g_host), (B, S, head, l1w)
    # raw C ABI: the host buffer may be overwritten as soon as the call has returned
    B, S, H = 2, 7, 24
    table = torch.stack([environment.scene_table(3, 4) for _ in range(B)])
    maps, tgt = _t(synth.make_maps(1, B, H), dev), _t(synth.make_maps(2, B, H), dev)
    ref_l, ref_g = native.rendering_loss(maps, tgt, table.to(dev))
    ws = torch.zeros(lib.svbrdf_rendering_loss_workspace_bytes(B, S, H, H) // 8 + 1, dtype=torch.int64, device=dev)
    loss, grad = torch.empty(1, device=dev), torch.empty_like(maps)
    xr = native.xrow(dev, H)
    host = np.ascontiguousarray(table.numpy()).copy()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.svbrdf_mixed_loss_fwd_bwd_host_scenes(maps.data_ptr(), tgt.data_ptr(), host.ctypes.data, xr.data_ptr(),
                                                   ctypes.c_float(0.1), ctypes.c_float(0.0), ctypes.c_float(0.01),
                                                   loss.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                                   B, S, H, H, st)
    host[:] = np.nan                                               # would poison a deferred read
    assert rc == 0, lib.svbrdf_last_error()
    torch.cuda.synchronize()
    assert torch.equal(loss, ref_l) and torch.equal(grad, ref_g)
    # one row too many: refused by the ABI ...
    big = torch.stack([environment.scene_table(144, 145) for _ in range(1)])
    m1, t1 = _t(synth.make_maps(3, 1, 8), dev), _t(synth.make_maps(4, 1, 8), dev)
    hb = np.ascontiguousarray(big.numpy())
    rc = lib.svbrdf_mixed_loss_fwd_bwd_host_scenes(m1.data_ptr(), t1.data_ptr(), hb.ctypes.data, native.xrow(dev, 8).data_ptr(),
                                                   ctypes.c_float(0.1), ctypes.c_float(0.0), ctypes.c_float(0.01),
                                                   loss.data_ptr(), None, ws.data_ptr(), ws.numel() * 8, 1, 289, 8, 8, st)
    assert rc == -2 and b"SVBRDF_HOST_SCENES_MAX_ROWS" in lib.svbrdf_last_error()
    # ... and uploaded by the wrapper
    l_up, g_up = native.rendering_loss(m1, t1, big)
    l_dv, g_dv = native.rendering_loss(m1, t1, big.to(dev))
    assert torch.equal(l_up, l_dv) and torch.equal(g_up, g_dv)
    # the C++ host path takes the same two routes
    from svbrdf_estimation_amd import _hostext
    ext = _hostext.module()
    assert ext is not None
    xa, xb = maps.clone().requires_grad_(True), maps.clone().requires_grad_(True)
    la = ext.fused_loss_with_scenes(xa, tgt, table, 0.1, 0.1, 0.01, st, False)
    lb = ext.fused_loss_with_scenes(xb, tgt, table.to(dev), 0.1, 0.1, 0.01, st, False)
    la.backward(); lb.backward()
    assert torch.equal(la, lb) and torch.equal(xa.grad, xb.grad)
    with pytest.raises(RuntimeError):
        ext.fused_loss_with_scenes(m1, t1, big, 0.1, 0.0, 0.01, st, False)
    # module level, a table too large for the argument block (2 x 150 rows): both host paths upload it
    from svbrdf_estimation_amd import losses, renderers
    res = []
    try:
        for enabled in (True, False):
            _hostext.set_enabled(enabled)
            fn = losses.RenderingLoss(renderers.LocalRenderer())
            fn.random_configuration_count, fn.specular_configuration_count = 50, 100
            x = maps.clone().requires_grad_(True)
            torch.manual_seed(11)
            l = fn(x, tgt)
            l.backward()
            res.append((l.detach().clone(), x.grad.clone()))
    finally:
        _hostext.set_enabled(True)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    torch.manual_seed(11)
    tab = torch.stack([environment.scene_table(50, 100) for _ in range(B)])
    l_ref, g_ref = native.rendering_loss(maps, tgt, tab.to(dev))
    assert torch.equal(res[0][0], l_ref.view(())) and torch.equal(res[0][1], g_ref)


# ---------------------------------------------------------------- row f1: network head fused into the loss

def test_head_fused_loss_golden(dev, native, oracle, golden):
    """models.py:338-346 head + MixedLoss / RenderingLoss in one kernel; gradient w.r.t. 9 channels"""
    from svbrdf_estimation_amd import _hostext, losses, renderers
    g = golden("g11_head_loss.npz")
    d_enc, d_tg, d_sc = _t(g["enc9"], dev), _t(g["target"], dev), _t(g["scenes"], dev)
    for tag, w in (("mixed", 0.1), ("render", 0.0)):
        loss, grad = native.rendering_loss(d_enc, d_tg, d_sc, l1_weight=w, head=True)
        ref_l, ref_g = oracle.head_loss(g["enc9"], g["target"], g["scenes"], w)
        _, g64 = oracle.head_loss(g["enc9"], g["target"], g["scenes"], w, f64=True)
        assert tuple(grad.shape) == tuple(g["enc9"].shape)
        assert_loss_close(loss.item(), ref_l, tag + " vs oracle")
        assert_loss_close(loss.item(), g[tag + "_loss"], tag + " vs reference", rtol=2e-6)
        assert_grad_close(_np(grad), ref_g, tag + " grad9 vs oracle", f64=g64)
        assert_grad_close(_np(grad), g[tag + "_grad9"], tag + " grad9 vs reference", f64=g64)
        # the unfused route -- decode_head in torch, then the 12-channel fused loss -- agrees
        x = d_enc.clone().requires_grad_(True)
        l2, _ = native.rendering_loss(losses.decode_head(x).detach(), d_tg, d_sc, l1_weight=w, want_grad=False)
        assert_loss_close(loss.item(), l2.item(), tag + " fused head vs torch decode", rtol=2e-6)
    np.testing.assert_allclose(_np(losses.decode_head(d_enc)), g["decoded12"], rtol=3e-7, atol=1e-7)
    # module interface, both host paths, same seed as the reference run
    res = []
    try:
        for enabled in (True, False):
            _hostext.set_enabled(enabled)
            x = d_enc.clone().requires_grad_(True)
            torch.manual_seed(int(g["rng_seed"]))
            loss = losses.FusedHeadLoss(renderers.LocalRenderer())(x, d_tg)
            loss.backward()
            res.append((loss.detach().clone(), x.grad.clone()))
    finally:
        _hostext.set_enabled(True)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert_loss_close(res[0][0].item(), g["mixed_loss"], "FusedHeadLoss module", rtol=2e-6)
    _, g64 = oracle.head_loss(g["enc9"], g["target"], g["scenes"], 0.1, f64=True)
    assert_grad_close(_np(res[0][1]), g["mixed_grad9"], "FusedHeadLoss grad", f64=g64)
    with pytest.raises(Exception):
        losses.FusedHeadLoss(renderers.LocalRenderer())(d_enc, d_tg.clone().requires_grad_(True)).backward()


# ---------------------------------------------------------------- row f4: training harness on the GPU

def test_training_harness_single_gpu_loss_decreases(dev, tmp_path):
    """train.py, one GPU: synthetic SVBRDFs, photos synthesised by K1, U-Net (stock), fused MixedLoss and
    the head-fused variant; a few Adam steps at a raised learning rate must lower the loss"""
    import train
    for extra in ([], ["--fused-head"]):
        args = train.parse_args(["--steps", "12", "--warmup", "2", "--batch", "2", "--workers", "0", "--lr", "2e-4",
                                 "--samples", "2"] + extra)
        res = train.run(args)
        assert np.isfinite(res["loss_last_quarter"]) and res["loss_last_quarter"] < res["loss_first_quarter"], res
    args = train.parse_args(["--model", "multi", "--views", "2", "--steps", "3", "--warmup", "1", "--batch", "1",
                             "--workers", "0", "--samples", "2"])
    assert np.isfinite(train.run(args)["loss_last_quarter"])


# ---------------------------------------------------------------- BASELINE configs[3] at its size

def test_config4_batch16_mixed_loss_module_path(dev, native, oracle):
    """configs[3]: batch 16, 256x256, 9 scenes, MixedLoss -- the loss shapes of the multi-view network's output.
    B*S = 144 scene rows ride in the launch's argument block like config 2's 72 (capacity 288); the device-table
    kernel is run on the same inputs as well and must agree bit for bit.  Checked at EVERY pixel of all 16 items
    against the C oracle, plus the size-independent properties (halves average to the whole, determinism, both host
    paths bitwise equal)."""
    from svbrdf_estimation_amd import _hostext, losses, renderers
    B, H, S = 16, 256, 9
    assert B * S <= native.host_scenes_max_rows()
    inp, tgt = synth.make_maps(1601, B, H), synth.make_maps(1602, B, H)
    d_in, d_tg = _t(inp, dev), _t(tgt, dev)
    loss_fn = losses.MixedLoss(renderers.LocalRenderer())
    torch.manual_seed(77)
    table = loss_fn.rendering_loss.sample_scene_table(B)
    res = []
    try:
        for enabled in (True, False):                  # native host extension, then python + ctypes
            _hostext.set_enabled(enabled)
            x = d_in.clone().requires_grad_(True)
            torch.manual_seed(77)
            loss = loss_fn(x, d_tg)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.item(), x.grad.clone()))
    finally:
        _hostext.set_enabled(True)
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    oracle.set_threads(min(32, oracle.max_threads()))
    ref_l, ref_g = oracle.mixed_loss(inp, tgt, table.numpy(), 0.1)
    assert_loss_close(res[0][0], ref_l, "config 4 mixed loss")
    _, g64 = oracle.mixed_loss(inp, tgt, table.numpy(), 0.1, f64=True)
    ties = oracle.loss_tie_map(inp, tgt, table.numpy())
    assert_grad_close(_np(res[0][1]), ref_g, "config 4 mixed-loss gradient", f64=g64, tie_map=ties,
                      max_ties=max(8, int(2e-6 * B * H * H * S * 3)))
    # halves: the batch loss is the mean of the two half-batch losses, the gradient of an item is 1/2 of its gradient
    # in a half-batch call (the mean's denominator)
    lo = native.rendering_loss(d_in[:8].contiguous(), d_tg[:8].contiguous(), table[:8].contiguous(), l1_weight=0.1)
    hi = native.rendering_loss(d_in[8:].contiguous(), d_tg[8:].contiguous(), table[8:].contiguous(), l1_weight=0.1)
    assert abs(0.5 * (lo[0].item() + hi[0].item()) - res[0][0]) <= 2e-7 * abs(res[0][0])
    halves = torch.cat((lo[1], hi[1]), dim=0) * 0.5
    assert_grad_close(_np(res[0][1]), _np(halves), "config 4 halves", rtol=2e-6, afrac=1e-7)
    # determinism of the device-table route (fixed-point loss reduction)
    again = native.rendering_loss(d_in, d_tg, native.upload_scene_table(table, dev), l1_weight=0.1)
    assert again[0].item() == res[0][0] and torch.equal(again[1], res[0][1])


@pytest.mark.timeout(1200)
def test_config4_training_harness_multi_view_5_batch_16(dev):
    """configs[3] through train.py: multi-view network (N = 5 photos, pooled encoder, models.py:348-411), batch 16,
    mixed loss, photos synthesised on the GPU -- two steps at size.  1 s with the in-tree MIOpen cache installed
    (tools/install_miopen_cache.sh; the driver's snapshot carries it), ~5 min of MIOpen kernel compilation on a box
    without it (the image ships no gfx950 database) -- slow then, not wrong."""
    import train
    args = train.parse_args(["--model", "multi", "--views", "5", "--batch", "16", "--steps", "1", "--warmup", "1",
                             "--workers", "0", "--samples", "16"])
    res = train.run(args)
    assert res["config"]["views"] == 5 and res["config"]["per_gpu_batch"] == 16
    assert np.isfinite(res["loss_first_quarter"]) and np.isfinite(res["loss_last_quarter"])
    print("config 4 end to end: %.1f patches/s (%.0f ms per step)" % (res["value"], res["ms_per_step"]))


# ---------------------------------------------------------------- rows f3 / f4 pinned against the reference

def _mix_restated_cpu(a, b, alpha):
    """test-side restatement of dataset.py:142-160 on the host, for shapes the fixture does not hold"""
    a, b, alpha = torch.as_tensor(a), torch.as_tensor(b), torch.as_tensor(alpha, dtype=torch.float32).view(-1, 1, 1, 1)
    n0, n1 = a[:, 0:3] / torch.clamp(a[:, 2:3], min=0.01), b[:, 0:3] / torch.clamp(b[:, 2:3], min=0.01)
    n = alpha * n0 + (1.0 - alpha) * n1
    n = n / torch.sqrt((n[:, 0:1] ** 2 + n[:, 1:2] ** 2) + n[:, 2:3] ** 2)
    return torch.cat((n, alpha * a[:, 3:] + (1.0 - alpha) * b[:, 3:]), dim=1).numpy()


def test_mix_materials_kernel_equals_reference_mix(dev, native, golden):
    """K4 against the reference's SvbrdfDataset.mix (dataset.py:142-160): explicit weight, weight drawn from the
    torch generator (same draw), the z < 0.01 projection floor, negative z; blended diffuse/roughness/specular bit
    for bit, normals within 2 ULP (torch's CPU sqrt is not correctly rounded)"""
    from svbrdf_estimation_amd import synthesis
    g = golden("g12_dataset_reader.npz")
    a, b = _t(g["mix__a"], dev), _t(g["mix__b"], dev)
    out = _np(synthesis.mix_materials(a, b, 0.3))
    assert np.array_equal(out[3:], g["mix__alpha03"][3:])
    assert np.abs(out[:3] - g["mix__alpha03"][:3]).max() <= 2.4e-7
    torch.manual_seed(int(g["mix__seed"]))
    out = _np(synthesis.mix_materials(a, b))
    assert np.array_equal(out[3:], g["mix__drawn"][3:]) and np.abs(out[:3] - g["mix__drawn"][:3]).max() <= 2.4e-7
    # batch with one weight per item, ragged sizes (every vector width), H != W
    for (B, H, W) in ((3, 7, 9), (2, 6, 10), (2, 16, 24), (1, 5, 5)):
        m0, m1 = synth.make_maps(710 + H, B, H, W, unit_normals=False), synth.make_maps(720 + W, B, H, W, tilt=0.8)
        m0[:, 2, 0, :] = np.float32(0.003)
        alpha = np.linspace(0.1, 0.9, B).astype(np.float32)
        got = _np(native.mix_materials(_t(m0, dev), _t(m1, dev), _t(alpha, dev)))
        ref = _mix_restated_cpu(m0, m1, alpha)
        assert np.array_equal(got[:, 3:], ref[:, 3:]), (B, H, W)
        assert np.abs(got[:, :3] - ref[:, :3]).max() <= 2.4e-7, (B, H, W)
    lib = native._load()
    p = a.data_ptr()
    assert lib.svbrdf_mix_materials(None, p, p, p, 1, 4, 4, None) == -1
    assert lib.svbrdf_mix_materials(p, p, p, p, 0, 4, 4, None) == -2


def test_mixing_dataset_item_through_the_gpu_mix(dev, golden, tmp_path):
    """SvbrdfDataset.__getitem__ with material mixing (dataset.py:52-56): dataset (host: partner + weight draws)
    -> collate -> apply_mixing (K4) reproduces the reference's mixed SVBRDF"""
    import random
    import shutil
    from svbrdf_estimation_amd.training import data
    g = golden("g12_dataset_reader.npz")
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for k in range(2):
        shutil.copy(os.path.join(gdir, "g12_maps_only_%d.png" % k), str(tmp_path))
    ds = data.TiledPngDataset(str(tmp_path), image_size=24, scale_mode="crop", image_count=0, used_image_count=0,
                              mix_materials=True)
    items = []
    for idx in (0, 1):
        random.seed(3 + idx)
        torch.manual_seed(21 + idx)
        items.append(ds[idx])
    batch = torch.utils.data.default_collate(items)
    mixed = _np(data.apply_mixing(batch["svbrdf"].to(dev), batch))
    for idx in (0, 1):
        ref = g["mixitem%d__svbrdf" % idx]
        assert np.array_equal(mixed[idx, 3:], ref[3:]) and np.abs(mixed[idx, :3] - ref[:3]).max() <= 2.4e-7


def test_mixing_dataset_with_resize_through_the_gpu(dev, golden, tmp_path):
    """scale_mode='resize' + mix_materials end to end: dataloader item (both materials centre-cropped, unresized) ->
    apply_mixing = K4 blend at full resolution, THEN bilinear resize on the device -- the reference's order
    (dataset.py:52-73) -- against the reference's __getitem__ output (g12 mixresize*)"""
    import random
    import shutil
    from svbrdf_estimation_amd.training import data
    g = golden("g12_dataset_reader.npz")
    for k in range(2):
        shutil.copy(os.path.join(os.path.dirname(__file__), "golden", "g12_maps_wide_%d.png" % k), str(tmp_path))
    ds = data.TiledPngDataset(str(tmp_path), image_size=20, image_count=0, used_image_count=0, scale_mode="resize",
                              mix_materials=True)
    items = []
    for idx in (0, 1):
        random.seed(13 + idx)
        torch.manual_seed(31 + idx)
        items.append(ds[idx])
    batch = torch.utils.data.default_collate(items)
    mixed = _np(data.apply_mixing(batch["svbrdf"].to(dev), batch))
    assert mixed.shape == (2, 12, 20, 20)
    for idx in (0, 1):
        err = np.abs(mixed[idx] - g["mixresize%d__svbrdf" % idx]).max()
        assert err <= 5e-7, (idx, err)


def test_uint8_transport_decoded_on_the_device_equals_the_float_reader(dev, tmp_path):
    """the 8-bit transport of the tiled-PNG reader: lookup-table decode ON THE DEVICE == the float path of the reader
    (which is pinned bit for bit against the reference's SvbrdfDataset), photos with gamma decode, normals, maps"""
    import shutil
    from svbrdf_estimation_amd.training import data
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    shutil.copy(os.path.join(gdir, "g12_tiled_toy_crop.png"), str(tmp_path))
    kw = dict(image_size=48, image_count=10, used_image_count=2, random_crop=True)
    a = data.TiledPngDataset(str(tmp_path), **kw)
    b = data.TiledPngDataset(str(tmp_path), uint8_transport=True, **kw)
    np.random.seed(11)
    fa = torch.utils.data.default_collate([a[0], a[0]])
    np.random.seed(11)
    fb = data.decode_uint8_batch(torch.utils.data.default_collate([b[0], b[0]]), dev)
    assert fb["inputs"].is_cuda and fb["inputs"].dtype == torch.float32
    assert torch.equal(fb["svbrdf"].cpu(), fa["svbrdf"])                     # value / 255 and * 2 - 1: exact arithmetic
    # the photos go through pow(x, 2.2), which torch's CPU kernels do not round correctly and not identically on their
    # vectorised and scalar paths: the 256-entry table (contiguous) and the reader's strided crop may differ in the last bit
    # on some hosts (they are equal, bit for bit, in the build container: tests/test_dataset_golden.py)
    err = (fb["inputs"].cpu() - fa["inputs"]).abs()
    assert (err <= 2.4e-7 * fa["inputs"].abs() + 1e-12).all(), err.max()


@pytest.mark.parametrize("tag", ["single", "multi"])
def test_unet_forward_equals_reference_fixture_on_the_gpu(dev, golden, tag):
    """the re-stated network on the MIOpen path against the reference's forward pass (regenerated weights, fixture
    g13): what test_training_models.py can only check where the reference is mounted"""
    from test_dataset_golden import unet_against_fixture
    err = unet_against_fixture(golden, tag, dev, rtol=1e-4, atol=1e-4)     # measured: 5e-6 / 1.2e-5 max abs error
    print("U-Net %s on the GPU vs reference fixture: max abs err %.2e" % (tag, err))


def test_plain_backward_is_one_launch_and_equals_the_autograd_engine(dev, golden):
    """`loss.backward()` on the native host path enters PyTorch's autograd engine from the extension
    (torch::autograd::backward) with the cached unit gradient: ONE kernel launch per step, and indistinguishable from
    ``torch.Tensor.backward(loss)``: same bits, same accumulation into an existing .grad, tensor hooks run, and every other
    use (explicit gradient, arithmetic on the loss, torch.autograd.grad, another stream, a target that needs a gradient, a
    second backward, retain_graph) behaves as plain autograd does.  (Until round 5 a leaf input took an engine-free
    shortcut that walked autograd internals; removed -- this test kept every equivalence it had pinned.)"""
    from svbrdf_estimation_amd import _hostext, losses, renderers
    assert _hostext.module() is not None
    g = golden("g3_loss_48.npz")
    d_in, d_tg = _t(g["input"], dev), _t(g["target"], dev)
    fn = losses.MixedLoss(renderers.LocalRenderer())

    def run(how, x=None, tg=d_tg):
        x = d_in.clone().requires_grad_(True) if x is None else x
        torch.manual_seed(5)
        loss = fn(x, tg)
        how(loss, x)
        torch.cuda.synchronize()
        return loss, x

    engine = run(lambda l, x: torch.Tensor.backward(l))                  # the plain autograd engine
    fast = run(lambda l, x: l.backward())
    assert isinstance(fast[0], losses._FusedLossTensor) and "_svbrdf_src" not in fast[0].__dict__    # consumed by backward()
    assert fast[0].item() == engine[0].item() and torch.equal(fast[1].grad, engine[1].grad)
    with pytest.raises(RuntimeError):
        fast[0].backward()                                               # already back-propagated: the engine says so
    # accumulation into an existing gradient
    x = d_in.clone().requires_grad_(True)
    x.grad = torch.ones_like(x)
    acc = run(lambda l, x: l.backward(), x=x)
    assert torch.equal(acc[1].grad, engine[1].grad + 1.0)
    # a tensor hook must run (engine path) and see the gradient
    seen = []
    x = d_in.clone().requires_grad_(True)
    x.register_hook(lambda gr: seen.append(gr.clone()))
    hooked = run(lambda l, x: l.backward(), x=x)
    assert len(seen) == 1 and torch.equal(seen[0], engine[1].grad) and torch.equal(hooked[1].grad, engine[1].grad)
    # explicit upstream gradient, arithmetic on the loss, torch.autograd.grad
    scaled = run(lambda l, x: l.backward(torch.tensor(2.0, device=dev)))
    assert torch.allclose(scaled[1].grad, engine[1].grad * 2.0, rtol=0, atol=0)
    arith = run(lambda l, x: (l * 3.0).backward())
    assert torch.equal(arith[1].grad, engine[1].grad * 3.0)
    x = d_in.clone().requires_grad_(True)
    torch.manual_seed(5)
    (ga,) = torch.autograd.grad(fn(x, d_tg), x)
    assert torch.equal(ga, engine[1].grad) and x.grad is None
    # backward issued from another stream than the forward: the engine inserts its stream sync
    side = torch.cuda.Stream(dev)
    def other_stream(l, x):
        with torch.cuda.stream(side):
            l.backward()
    moved = run(other_stream)
    assert torch.equal(moved[1].grad, engine[1].grad)
    # a target that needs its gradient too
    tg = d_tg.clone().requires_grad_(True)
    both = run(lambda l, x: l.backward(), tg=tg)
    assert both[0].__dict__.get("_svbrdf_src") is None                  # consumed by backward()
    assert torch.equal(both[1].grad, engine[1].grad) and tg.grad is not None and bool(tg.grad.abs().sum() > 0)
    # a non-leaf input (the training case: the maps come out of a network): its plain backward() goes through the engine
    # with the extension's cached unit gradient (no fill kernel, no scale launch) and must give the bits the engine gives
    # with its own ones tensor -- also under retain_graph, and a second time
    ws = []
    for mode in ("engine", "unit", "unit_python_front_end", "unit_retain"):
        w = torch.ones(1, device=dev, requires_grad=True)
        torch.manual_seed(5)
        loss = fn(d_in * w, d_tg)
        assert isinstance(loss, losses._FusedLossTensor) and "_svbrdf_src" in loss.__dict__
        if mode == "engine":
            torch.Tensor.backward(loss)
        elif mode == "unit":                                             # the engine entered from the extension (C++ API)
            loss.backward()
            with pytest.raises(RuntimeError):
                loss.backward()                                          # freed graph: autograd says so
        elif mode == "unit_python_front_end":                            # ... and through torch.autograd.backward
            try:
                losses._ENGINE_FROM_NATIVE = False
                loss.backward()
            finally:
                losses._ENGINE_FROM_NATIVE = True
        else:
            loss.backward(retain_graph=True)
            first = w.grad.clone()
            loss.backward()                                              # second pass: the engine's own ones, scaled copy
            assert torch.equal(w.grad, first * 2.0)
            w.grad = first
        torch.cuda.synchronize()
        ws.append(w.grad.clone())
    assert torch.isfinite(ws[0]).all() and all(torch.equal(ws[0], other) for other in ws[1:])
    # errors raised inside the graph surface as Python exceptions through the native entry too (a Python-defined node that
    # raises, below the loss)
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise ValueError("boom from a python node")
    torch.manual_seed(5)
    loss = fn(Boom.apply(d_in.clone().requires_grad_(True)), d_tg)
    with pytest.raises((ValueError, RuntimeError), match="boom"):
        loss.backward()
    # the unit gradient is recognised by ADDRESS: an equal-valued other tensor takes the scaling path, same result
    x = d_in.clone().requires_grad_(True)
    other_one = run(lambda l, x: l.backward(torch.ones((), device=dev)), x=x)
    assert torch.equal(other_one[1].grad, engine[1].grad)
    ext = _hostext.module()
    u = ext.unit_gradient(engine[0].detach())
    assert u.item() == 1.0 and u.data_ptr() == ext.unit_gradient(engine[0].detach()).data_ptr() and not u.requires_grad
    # ... and by VERSION: a loss-scaling hook that edits the incoming gradient in place keeps the address.  A hooked loss is
    # given the engine's own ones tensor (the hook may do what it likes with it); the cached 1.0 itself, edited in place by
    # whoever got hold of it, is not believed (the scale is applied) and is replaced by a fresh 1.0 for the next backward
    from svbrdf_estimation_amd import _native as native

    def hooked(scale_in_place):
        w = torch.ones(1, device=dev, requires_grad=True)
        torch.manual_seed(5)
        loss = fn(d_in * w, d_tg)
        loss.register_hook((lambda g: g.mul_(3.0)) if scale_in_place else (lambda g: g * 3.0))
        before = native.launch_count()
        loss.backward()
        torch.cuda.synchronize()
        return w.grad.clone(), native.launch_count() - before
    for in_place in (True, False):
        got, launches = hooked(in_place)
        assert torch.allclose(got, ws[0] * 3.0, rtol=1e-6, atol=0) and launches == 1, (in_place, got, ws[0], launches)   # the scale launch
    w = torch.ones(1, device=dev, requires_grad=True)
    torch.manual_seed(5)
    loss = fn(d_in * w, d_tg)
    u = ext.unit_gradient(loss.detach())
    u.mul_(5.0)                                            # someone scribbles on the cached tensor ...
    fresh = ext.unit_gradient(loss.detach())
    assert fresh.item() == 1.0 and fresh.data_ptr() != u.data_ptr()          # ... it is replaced, not trusted
    torch.Tensor.backward(loss, u)                         # and the scribbled one, handed in explicitly, is an ordinary gradient
    torch.cuda.synchronize()
    assert torch.allclose(w.grad, ws[0] * 5.0, rtol=1e-6, atol=0)
    w2 = torch.ones(1, device=dev, requires_grad=True)
    torch.manual_seed(5)
    fn(d_in * w2, d_tg).backward()                         # the next plain backward is exact again
    torch.cuda.synchronize()
    assert torch.equal(w2.grad, ws[0])
    # switched off, a plain backward is the engine's with its own ones-fill and the node's scale launch (bench.py's
    # three-launch reference leg); a plain backward with it on: no launch at all beyond the forward's
    from svbrdf_estimation_amd import _native as native
    try:
        losses._UNIT_GRADIENT = False
        before = native.launch_count()
        plain = run(lambda l, x: l.backward())
        assert native.launch_count() - before == 2                      # K3 + the scale
    finally:
        losses._UNIT_GRADIENT = True
    assert torch.equal(plain[1].grad, engine[1].grad)
    before = native.launch_count()
    run(lambda l, x: l.backward())
    assert native.launch_count() - before == 1                          # K3 only


# ---------------------------------------------------------------- streams

def test_non_default_and_concurrent_streams(dev, native, golden):
    """kernels are enqueued on torch's current stream; calls on different streams use separate loss scratch"""
    from svbrdf_estimation_amd import _hostext, losses, renderers
    g = golden("g3_loss_48.npz")
    d_in, d_tg, d_sc = _t(g["input"], dev), _t(g["target"], dev), _t(g["scenes"], dev)
    ref_l, ref_g = native.rendering_loss(d_in, d_tg, d_sc)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = {}
    for rep in range(5):                       # interleave launches on the two streams
        for name, st in (("a", s1), ("b", s2)):
            with torch.cuda.stream(st):
                outs[name] = native.rendering_loss(d_in, d_tg, d_sc)
    torch.cuda.synchronize()
    for name in ("a", "b"):
        assert outs[name][0].item() == ref_l.item() and torch.equal(outs[name][1], ref_g)
    # module path (native host extension and ctypes) on a side stream, seed-reproducible
    res = []
    try:
        for enabled in (True, False):
            _hostext.set_enabled(enabled)
            with torch.cuda.stream(s1):
                x = d_in.clone().requires_grad_(True)
                torch.manual_seed(int(g["rng_seed"]))
                loss = losses.RenderingLoss(renderers.LocalRenderer())(x, d_tg)
                loss.backward()
            s1.synchronize()
            res.append((loss.item(), x.grad.clone()))
    finally:
        _hostext.set_enabled(True)
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    # (the scenes are re-sampled on this host: equal to the fixture's table up to the CPU math library)
    assert_loss_close(res[0][0], ref_l.item(), "side-stream module loss", rtol=2e-6)
