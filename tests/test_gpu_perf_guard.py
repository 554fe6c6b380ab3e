"""GPU-side speed guard: the kernels' measured speed, through the C ABI, must not regress.

tests/test_isa_guard.py pins instruction counts at compile time; this file pins what those counts are for -- time on
the device -- so that a change which keeps the counts but loses the schedule, the occupancy or the memory pattern
turns the GPU suite red instead of silently costing 15 %.

How one figure is measured (``measure``):
  * the exported symbols are called with raw device pointers FROM C (tests/c_host/launch_loop.c, built with gcc on the
    spot, takes the entry point as a function pointer: the host side of a launch is the ABI function itself, a few
    microseconds; issued through Python / ctypes a slow host needs 20-35 us per launch, as long as K3 runs), on a
    stream of their own;
  * the timed launches follow an untimed prefill of as many on the same stream, so the queue is deep when the region
    opens and the events around it span the device running launches back to back (checked: the region is enqueued in
    four chunks, and at every chunk boundary the device has not yet reached the mark set one chunk earlier);
  * the VALU-bound loss kernel is judged in SHADER CYCLES, not microseconds: boxes of this pool hold 1.94-2.18 GHz
    under this kernel (profiles/r04_k3_clock_ab.txt), a spread wider than any regression worth catching.  A second
    probe, on a second stream, opens with the timed launches (it waits for the region's first event), spins for ~90 %
    of them and reads the clock the chip holds under exactly this load (s_memtime / s_memrealtime);
  * the maps rotate over several sets (432 MiB at config 2: beyond the 256 MiB Infinity Cache), as in bench.py;
  * best of three repeats (a guard asks "can the kernel still do it", not "what does it do on average").

Thresholds and where they come from (cycles = time per launch x clock under that load; measured by this harness on
five boxes of rounds 5 and 6, profiles/r05_perf_guard.json, profiles/r06_perf_guard.json; they agree within 1.5 %):
  K3 config 2 (B = 8, 256x256, 9 scenes, tied roughness)   <= 90,000 cycles per launch      measured 85.5-87.6 k
  K3 config 2, MixedLoss (the training loss)               <= 95,000                        measured 91.2-92.1 k
  K3 config 2, untied roughness (three lobes)              <= 126,000                       measured 119.2-122.1 k
  K3 config-5 shape (B = 8, 512x512, 11 + 21 scenes)       <= 930,000                       measured 893.9-895.9 k
        = the slowest box seen + 3 % (round 5 shipped + 7-8 %; the round-5 review asked for what five agreeing boxes
        support and proposed 124 k for the untied shape from one box's 119.3 k -- two other boxes read 121.9 and 122.1 k, so
        that bound would have been 1.5 % away from a healthy box).  The
        repeat that is kept is the one with the FEWEST CYCLES (clock and duration of the same repeat), every repeat is
        recorded in perf_guard.json.
        NOTE on cycles: the clock under this kernel is not one number -- it moves between 2.0 and 2.4 GHz within
        milliseconds, differs by box, and sags to ~1.75 GHz for ~5 ms when load follows an idle period
        (tools/clock_timeline.py, profiles/r05_clock_timeline*.txt).  Here the probe opens with the timed launches (same
        event) and spans ~90 % of them; bench.py pairs clock and duration of one interval since round 5.
  K1 / K2 (288 renders of 256x256, one per map) and K4 (64 samples) >= 0.72 of 8 TB/s
        BENCH_r05.json: 0.841 / 0.792 / 0.780; this harness: 0.834 / 0.810 / 0.773.  HBM-bound: judged in bytes per
        second (the HBM clock is not the shader clock).
  svbrdf_debug_copy (1 GiB -> 1 GiB, beyond the Infinity Cache)       >= 0.72 of 8 TB/s
        the copy bandwidth of the box (MI355X_MICROARCH.md: 6.29 TB/s = 0.79): what "HBM-bound" can reach here, and the
        denominator of bench.py's frac_of_measured_copy_peak.
  K1 + sensor noise + clamp (svbrdf_render_inputs, 288 photos of 256x256, one per map): recorded, and guarded against
        the three-pass form it replaces (K1, then torch's randn / multiply-add / clamp passes): >= 1.5x faster.
  one kernel launch per training step of the loss (svbrdf_debug_launch_count around 64 steps on a NON-LEAF input).

Run as a script (``python tests/test_gpu_perf_guard.py``) it prints the measurements as JSON and writes
gpurun_out/perf_guard.json -- how the thresholds were obtained.
"""
import ctypes
import json
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

HBM_PEAK = 8.0e12

# ---- thresholds (see the module docstring) ------------------------------------------------------------------------
K3_CONFIG2_MAX_CYCLES = 90_000
K3_MIXED_MAX_CYCLES = 95_000
K3_UNTIED_MAX_CYCLES = 126_000
K3_CONFIG5_MAX_CYCLES = 930_000
HBM_KERNELS_MIN_FRAC = 0.72
COPY_MIN_FRAC = 0.72
FUSED_PHOTOS_MIN_SPEEDUP = 1.5

_fp = ctypes.c_void_p


def _maps(gen, B, H, tied=True):
    from bench import synthetic_maps
    return synthetic_maps(gen, B, H, tied=tied)


def _build_launch_loops():
    """tests/c_host/launch_loop.c -> a temporary shared object (gcc): the launches of a measured region are issued from C"""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="svbrdf_perf_guard_"), "launch_loop.so")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tests", "c_host", "launch_loop.c")])
    return ctypes.CDLL(out)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


class Harness:
    def __init__(self, dev):
        from svbrdf_estimation_amd import _native
        self.native = _native
        self.lib = _native._load()
        self.loops = _build_launch_loops()
        for name in ("perf_loop_loss", "perf_loop_render_fwd", "perf_loop_render_bwd", "perf_loop_mix", "perf_loop_copy",
                     "perf_loop_render_inputs"):
            getattr(self.loops, name).restype = ctypes.c_int
        self.dev = dev
        self.sa, self.sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        self.probe = self.lib.svbrdf_debug_clock_probe
        self.probe.argtypes = [_fp, ctypes.c_ulonglong, _fp]
        self.probe.restype = ctypes.c_int
        self.clk_out = torch.zeros(2, dtype=torch.int64, device=dev)
        assert self.probe(self.clk_out.data_ptr(), 1, _fp(self.sb.cuda_stream)) == 0     # first use loads the kernel (~6 ms of
        torch.cuda.synchronize(dev)                                                        # host time): not inside a region

    def _fn(self, name):
        """the entry point of the library under test as a plain function pointer for the C loops"""
        return ctypes.cast(getattr(self.lib, name), ctypes.c_void_p)

    def measure(self, enqueue, est_us, want_clock=True, repeats=3):
        """enqueue(first, n): enqueues launches first .. first+n-1 on stream ``self.sa`` (from C: tests/c_host/launch_loop.c).
        -> dict(us_per_launch, clock_GHz, cycles_per_launch, repeats=[every valid repeat]): the best repeat -- by CYCLES when
        the clock is read (the assertion is on cycles, and the chip lowers its clock for denser issue: the fastest repeat is
        not always the one with the fewest cycles), by microseconds otherwise."""
        dev, sa, sb = self.dev, self.sa, self.sb
        n = 4 * int(max(3, min(30, 1000.0 / est_us)))       # ~4 ms of launches, in four chunks
        q = n // 4
        probe_ticks = int(max(20000, min(600000, 0.9 * n * est_us * 100)))   # ~90 % of the region, in 10 ns ticks
        # settle: clocks and caches in their steady state
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            enqueue(0, 32)
            torch.cuda.synchronize(dev)
        best, invalid, valid, every = None, [], 0, []
        for _ in range(repeats + 2):            # up to two attempts may be spoilt by the host
            e0 = torch.cuda.Event(enable_timing=True)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            torch.cuda.synchronize(dev)
            enqueue(0, n)                                # untimed prefill: the queue is deep before the region opens
            e0.record(sa)
            if want_clock:
                sb.wait_event(e0)
                assert self.probe(self.clk_out.data_ptr(), probe_ticks, _fp(sb.cuda_stream)) == 0
            t_host = time.perf_counter()
            behind = []
            for c in range(4):
                enqueue(n + c * q, q)
                marks[c].record(sa)
                if c:       # has the device already passed the mark set one chunk ago?  Then it is about to run dry.
                    behind.append(not marks[c - 1].query())
            host_ms = 1e3 * (time.perf_counter() - t_host)
            torch.cuda.synchronize(dev)
            dev_ms = e0.elapsed_time(marks[-1])
            # The region must be the device running launches back to back, whatever the host does: at every chunk
            # boundary the device still had at least one whole chunk (~1 ms of launches) queued in front of it.  (The
            # host's enqueue time says nothing: with ~200 launches queued -- 10 KB of kernel arguments each -- the launch
            # call blocks until the device frees a slot, and the host then reads exactly as slow as the device.)
            if not all(behind):         # not a measurement of the kernel: try again (a hiccup of the host), fail below if it persists
                invalid.append("the device caught up with the host inside the region: %s (host %.2f ms, device %.2f ms)" % (behind, host_ms, dev_ms))
                continue
            us = 1e3 * dev_ms / n
            ghz = None
            if want_clock:
                cyc, ticks = (int(v) for v in self.clk_out.tolist())
                ghz = cyc / ticks * 0.1
            res = {"us_per_launch": us, "launches": n, "clock_GHz": ghz,
                   "cycles_per_launch": us * ghz * 1e3 if ghz else None, "host_enqueue_ms": host_ms}
            every.append(dict(res))
            key = "cycles_per_launch" if want_clock else "us_per_launch"
            if best is None or res[key] < best[key]:
                best = res
            valid += 1
            if valid >= repeats:
                break
        assert best is not None, "no valid measurement in %d attempts: %s" % (repeats + 2, invalid)
        best = dict(best)
        best["repeats"] = every
        return best

    # ---- the cases ------------------------------------------------------------------------------------------
    def k3_calls(self, B, H, n_random, n_specular, tied=True, l1_weight=0.0, sets=6):
        from svbrdf_estimation_amd import environment
        lib, dev = self.lib, self.dev
        gen = torch.Generator().manual_seed(5)
        S = n_random + n_specular
        torch.manual_seed(11)
        table = environment.BatchSceneSampler(B, n_random, n_specular).sample().contiguous()
        assert B * S <= self.native.host_scenes_max_rows()
        xr = self.native.xrow(dev, H)
        nbytes = lib.svbrdf_rendering_loss_workspace_bytes(B, S, H, H)
        ws = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        loss = torch.empty(1, device=dev)
        ins = [_maps(gen, B, H, tied).to(dev) for _ in range(sets)]
        tgs = [_maps(gen, B, H, tied).to(dev) for _ in range(sets)]
        grads = [torch.empty_like(a) for a in ins]
        p_in, p_tg, p_gr = _ptr_array(ins), _ptr_array(tgs), _ptr_array(grads)
        fn, loop, st = self._fn("svbrdf_mixed_loss_fwd_bwd_host_scenes"), self.loops.perf_loop_loss, _fp(self.sa.cuda_stream)
        c_f, c_i = ctypes.c_float, ctypes.c_int

        def enqueue(first, n):
            rc = loop(fn, c_i(n), c_i(first), c_i(sets), p_in, p_tg, p_gr, _fp(table.data_ptr()), _fp(xr.data_ptr()), c_f(0.1),
                      c_f(l1_weight), c_f(0.01), _fp(loss.data_ptr()), _fp(ws.data_ptr()), ctypes.c_size_t(ws.numel() * 8),
                      c_i(B), c_i(S), c_i(H), c_i(H), st)
            assert rc == 0, lib.svbrdf_last_error()
        torch.cuda.synchronize(dev)
        return enqueue, [table, xr, ws, loss, ins, tgs, grads, p_in, p_tg, p_gr], 144.0 * H * H * B

    def k12_calls(self, which):
        from svbrdf_estimation_amd import environment
        lib, dev = self.lib, self.dev
        B, H = 288, 256
        gen = torch.Generator().manual_seed(7)
        maps = _maps(gen, B, H).to(dev)
        torch.manual_seed(7)
        table = environment.BatchSceneSampler(B, 1, 0).sample().to(dev)
        xr = self.native.xrow(dev, H)
        st, c_i = _fp(self.sa.cuda_stream), ctypes.c_int
        if which == "k1":
            out = torch.empty(B, 1, 3, H, H, device=dev)
            fn, nbytes, keep = self._fn("svbrdf_render_fwd"), 60.0 * H * H * B, [maps, table, xr, out]

            def enqueue(first, n):
                rc = self.loops.perf_loop_render_fwd(fn, c_i(n), _fp(maps.data_ptr()), _fp(table.data_ptr()), _fp(xr.data_ptr()),
                                                     _fp(out.data_ptr()), c_i(B), c_i(1), c_i(H), c_i(H), st)
                assert rc == 0, lib.svbrdf_last_error()
        else:
            cot = torch.randn(B, 1, 3, H, H, device=dev)
            out = torch.empty(B, 12, H, H, device=dev)
            fn, nbytes, keep = self._fn("svbrdf_render_bwd"), 108.0 * H * H * B, [maps, table, xr, cot, out]

            def enqueue(first, n):
                rc = self.loops.perf_loop_render_bwd(fn, c_i(n), _fp(maps.data_ptr()), _fp(table.data_ptr()), _fp(xr.data_ptr()),
                                                     _fp(cot.data_ptr()), _fp(out.data_ptr()), c_i(B), c_i(1), c_i(H), c_i(H), st)
                assert rc == 0, lib.svbrdf_last_error()
        torch.cuda.synchronize(dev)
        return enqueue, keep, nbytes

    def k4_calls(self):
        lib, dev = self.lib, self.dev
        B, H, sets = 64, 256, 2     # two alternating sets (604 MB), as in bench.py / tools/kernel_cases.py
        gen = torch.Generator().manual_seed(17)
        alpha = torch.rand(B, device=dev) * 0.8 + 0.1
        a = [_maps(gen, B, H).to(dev) for _ in range(sets)]
        b = [_maps(gen, B, H).to(dev) for _ in range(sets)]
        outs = [torch.empty_like(x) for x in a]
        p_a, p_b, p_o = _ptr_array(a), _ptr_array(b), _ptr_array(outs)
        fn, st, c_i = self._fn("svbrdf_mix_materials"), _fp(self.sa.cuda_stream), ctypes.c_int

        def enqueue(first, n):
            rc = self.loops.perf_loop_mix(fn, c_i(n), c_i(first), c_i(sets), p_a, p_b, _fp(alpha.data_ptr()), p_o, c_i(B), c_i(H),
                                          c_i(H), st)
            assert rc == 0, lib.svbrdf_last_error()
        torch.cuda.synchronize(dev)
        return enqueue, [alpha, a, b, outs, p_a, p_b, p_o], 144.0 * H * H * B


    def copy_calls(self, gib=1.0):
        dev = self.dev
        n = int(gib * 2 ** 30) // 4                      # floats; src + dst = 2 GiB: far beyond the 256 MiB Infinity Cache
        src = torch.empty(n, device=dev).uniform_(-1.0, 1.0)
        dst = torch.empty_like(src)
        fn, st = self._fn("svbrdf_debug_copy"), _fp(self.sa.cuda_stream)

        def enqueue(first, k):
            rc = self.loops.perf_loop_copy(fn, ctypes.c_int(k), _fp(dst.data_ptr()), _fp(src.data_ptr()), ctypes.c_size_t(n), st)
            assert rc == 0, self.lib.svbrdf_last_error()
        torch.cuda.synchronize(dev)
        return enqueue, [src, dst], 8.0 * n

    def photo_calls(self):
        """svbrdf_render_inputs, device tables: 288 maps, one photo each (the HBM-heaviest shape: 60 B per pixel)"""
        from svbrdf_estimation_amd import synthesis
        dev = self.dev
        B, H = 288, 256
        gen = torch.Generator().manual_seed(7)
        maps = _maps(gen, B, H).to(dev)
        torch.manual_seed(7)
        table = torch.stack([synthesis.input_scene_table(1, True) for _ in range(B)]).to(dev)
        levels = synthesis.noise_levels(B).to(dev)
        xr = self.native.xrow(dev, H)
        out = torch.empty(B, 1, 3, H, H, device=dev)
        fn, st, c_i = self._fn("svbrdf_render_inputs"), _fp(self.sa.cuda_stream), ctypes.c_int

        def enqueue(first, n):
            rc = self.loops.perf_loop_render_inputs(fn, c_i(n), c_i(first), _fp(maps.data_ptr()), _fp(table.data_ptr()),
                                                    _fp(levels.data_ptr()), ctypes.c_ulonglong(99), _fp(xr.data_ptr()),
                                                    _fp(out.data_ptr()), c_i(B), c_i(1), c_i(H), c_i(H), st)
            assert rc == 0, self.lib.svbrdf_last_error()
        torch.cuda.synchronize(dev)
        return enqueue, [maps, table, levels, xr, out], 60.0 * H * H * B


@pytest.fixture(scope="module")
def harness():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (select CPU tests with -m 'not gpu')"
    return Harness(torch.device("cuda:0"))


_RESULTS = {}


def _flush():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "perf_guard.json"), "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "results": _RESULTS}, f, indent=1)


def _record(name, res, nbytes):
    res = dict(res)
    res["algorithmic_bytes_per_launch"] = nbytes
    res["frac_of_hbm_peak"] = nbytes / (res["us_per_launch"] * 1e-6) / HBM_PEAK
    _RESULTS[name] = res
    _flush()
    print("[perf-guard] %s: %.2f us per launch, %s cycles at %s GHz, %.3f of 8 TB/s" % (
        name, res["us_per_launch"], "%.0f" % res["cycles_per_launch"] if res["cycles_per_launch"] else "-",
        "%.3f" % res["clock_GHz"] if res["clock_GHz"] else "-", res["frac_of_hbm_peak"]))
    return res


def test_k3_config2_cycles(harness):
    calls, keep, nbytes = harness.k3_calls(8, 256, 3, 6)
    res = _record("k3_config2", harness.measure(calls, est_us=38.0), nbytes)
    assert 1.5 < res["clock_GHz"] < 2.6, res
    assert res["cycles_per_launch"] <= K3_CONFIG2_MAX_CYCLES, res


def test_k3_mixed_loss_cycles(harness):
    calls, keep, nbytes = harness.k3_calls(8, 256, 3, 6, l1_weight=0.1)
    res = _record("k3_config2_mixed", harness.measure(calls, est_us=40.0), nbytes)
    assert res["cycles_per_launch"] <= K3_MIXED_MAX_CYCLES, res


def test_k3_untied_cycles(harness):
    calls, keep, nbytes = harness.k3_calls(8, 256, 3, 6, tied=False)
    res = _record("k3_config2_untied", harness.measure(calls, est_us=52.0), nbytes)
    assert res["cycles_per_launch"] <= K3_UNTIED_MAX_CYCLES, res


def test_k3_config5_shape_cycles(harness):
    calls, keep, nbytes = harness.k3_calls(8, 512, 11, 21, sets=2)
    res = _record("k3_config5_shape", harness.measure(calls, est_us=400.0), nbytes)
    assert res["cycles_per_launch"] <= K3_CONFIG5_MAX_CYCLES, res


@pytest.mark.parametrize("which", ["k1", "k2"])
def test_render_kernels_hbm_fraction(harness, which):
    calls, keep, nbytes = harness.k12_calls(which)
    res = _record(which + "_288_renders", harness.measure(calls, est_us=170.0 if which == "k1" else 325.0, want_clock=False),
                  nbytes)
    assert res["frac_of_hbm_peak"] >= HBM_KERNELS_MIN_FRAC, res


def test_mix_materials_hbm_fraction(harness):
    calls, keep, nbytes = harness.k4_calls()
    res = _record("k4_64_samples", harness.measure(calls, est_us=100.0, want_clock=False), nbytes)
    assert res["frac_of_hbm_peak"] >= HBM_KERNELS_MIN_FRAC, res


def test_copy_kernel_reaches_the_copy_peak(harness):
    """svbrdf_debug_copy on 1 GiB -> 1 GiB: the measured-copy peak of this box"""
    calls, keep, nbytes = harness.copy_calls()
    res = _record("copy_1GiB", harness.measure(calls, est_us=340.0, want_clock=False), nbytes)
    assert res["frac_of_hbm_peak"] >= COPY_MIN_FRAC, res
    src, dst = keep
    assert torch.equal(src, dst)


def test_fused_photo_synthesis_beats_the_three_pass_form(harness):
    """K1 + noise + clamp in one launch (svbrdf_render_inputs) against what it replaces: K1, then torch's randn_like,
    multiply-add and clamp passes over the photos (round 5's synthesis.render_inputs)."""
    calls, keep, nbytes = harness.photo_calls()
    res = _record("k1_noise_clamp_288_photos", harness.measure(calls, est_us=200.0, want_clock=False), nbytes)
    maps, table, levels, xr, out = keep
    dev = harness.dev
    sig = levels.view(-1, 1, 1, 1, 1)

    def three_pass():
        o = harness.native.render_fwd(maps, table)
        o = o + torch.randn_like(o) * sig
        return o.clamp_(0.0, 1.0)
    for _ in range(3):
        three_pass()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        three_pass()
    e1.record()
    torch.cuda.synchronize(dev)
    old_us = 1e3 * e0.elapsed_time(e1) / 10
    _RESULTS["k1_noise_clamp_288_photos"]["three_pass_form_us"] = old_us
    _RESULTS["k1_noise_clamp_288_photos"]["speedup_over_three_pass_form"] = old_us / res["us_per_launch"]
    _flush()
    print("[perf-guard] photos: fused %.1f us, K1 + randn + fma + clamp passes %.1f us" % (res["us_per_launch"], old_us))
    assert old_us >= FUSED_PHOTOS_MIN_SPEEDUP * res["us_per_launch"], (old_us, res)


def test_one_kernel_launch_per_training_step():
    """values were tested bit-identical for the one-launch engine path, the launch COUNT was
    asserted nowhere.  64 steps of ``loss.backward()`` on a NON-LEAF input (a network output: PyTorch's autograd engine
    runs the node) enqueue exactly 64 kernels of this library -- no scale launch -- and, seen by torch's profiler, the
    device ran exactly one kernel per step that is not the stand-in network's own: no ones-fill by the engine."""
    from svbrdf_estimation_amd import _native, losses, renderers
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    B, H, steps = 8, 256, 64
    x = _maps(gen, B, H).to(dev).requires_grad_(True)
    tgt = _maps(gen, B, H).to(dev)
    for name, fn in (("RenderingLoss", losses.RenderingLoss(renderers.LocalRenderer())),
                     ("MixedLoss", losses.MixedLoss(renderers.LocalRenderer()))):
        def step():
            x.grad = None
            y = x * 1.0                    # a non-leaf: what a network output is to the loss
            fn(y, tgt).backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        before = _native.launch_count()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        assert _native.launch_count() - before == steps, (name, _native.launch_count() - before)
        assert x.grad is not None and torch.isfinite(x.grad).all()
        # the whole device timeline of 16 steps: per step the stand-in's multiply, its backward (multiply) and ONE loss kernel
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(16):
                step()
            torch.cuda.synchronize(dev)
        dev_type = getattr(torch.autograd, "DeviceType", None)
        kernels = [e for e in prof.events() if dev_type is not None and e.device_type == dev_type.CUDA]
        names = [e.name for e in kernels]
        if names:           # kineto saw the device (it does on ROCm builds with roctracer; an empty trace proves nothing)
            ours = [n for n in names if "k_rendering_loss" in n]
            fills = [n for n in names if "fill" in n.lower() or "k_scale_inplace" in n]
            _RESULTS["launches_per_step_" + name] = {"device_kernels_in_16_steps": len(names), "loss_kernels": len(ours),
                                                     "fill_or_scale_kernels": len(fills), "distinct": sorted(set(names))[:12]}
            assert len(ours) == 16 and not fills, (name, sorted(set(names)))
        else:
            _RESULTS["launches_per_step_" + name] = {"device_kernels_in_16_steps": None,
                                                     "note": "torch.profiler reported no device events on this build"}
        _flush()
        print("[perf-guard] %s: %s" % (name, _RESULTS["launches_per_step_" + name]))


if __name__ == "__main__":
    h = Harness(torch.device("cuda:0"))
    for name, (mk, est, clk) in {
            "k3_config2": (lambda: h.k3_calls(8, 256, 3, 6), 38.0, True),
            "k3_config2_untied": (lambda: h.k3_calls(8, 256, 3, 6, tied=False), 52.0, True),
            "k3_config2_mixed": (lambda: h.k3_calls(8, 256, 3, 6, l1_weight=0.1), 42.0, True),
            "k3_config5_shape": (lambda: h.k3_calls(8, 512, 11, 21, sets=2), 400.0, True),
            "k1_288_renders": (lambda: h.k12_calls("k1"), 170.0, False),
            "k2_288_renders": (lambda: h.k12_calls("k2"), 325.0, False),
            "k4_64_samples": (lambda: h.k4_calls(), 100.0, False),
            "copy_1GiB": (lambda: h.copy_calls(), 340.0, False),
            "k1_noise_clamp_288_photos": (lambda: h.photo_calls(), 200.0, False)}.items():
        if os.environ.get("PERF_GUARD_CASES") and name not in os.environ["PERF_GUARD_CASES"].split(","):
            continue
        calls, keep, nbytes = mk()
        for rep in range(int(os.environ.get("PERF_GUARD_REPEATS", "2"))):
            _record(name if rep == 0 else "%s_run%d" % (name, rep + 1), h.measure(calls, est, want_clock=clk), nbytes)
        del calls, keep
        torch.cuda.empty_cache()
    print(json.dumps(_RESULTS))
