"""GPU-side speed guard: the kernels' measured speed, through the C ABI, must not regress.

tests/test_isa_guard.py pins instruction counts at compile time; this file pins what those counts are for -- time on
the device -- so that a change which keeps the counts but loses the schedule, the occupancy or the memory pattern
turns the GPU suite red instead of silently costing 15 %.

How one figure is measured (``measure``):
  * the exported symbols are called with raw device pointers (ctypes, argument tuples built once: ~5 us of host time
    per launch, far below every kernel here), on a stream of their own;
  * the timed launches follow an untimed prefill of as many on the same stream, so the queue is deep when the region
    opens and the event pair around it spans the device running launches back to back (checked: the host needs less
    time to enqueue the region's launches than the device to run them);
  * the VALU-bound loss kernel is judged in SHADER CYCLES, not microseconds: boxes of this pool hold 1.94-2.18 GHz
    under this kernel (profiles/r04_k3_clock_ab.txt), a spread wider than any regression worth catching.  A second
    probe, on a second stream, opens with the timed launches (it waits for the region's first event), spins for ~90 %
    of them and reads the clock the chip holds under exactly this load (s_memtime / s_memrealtime);
  * the maps rotate over several sets (432 MiB at config 2: beyond the 256 MiB Infinity Cache), as in bench.py;
  * best of three repeats (a guard asks "can the kernel still do it", not "what does it do on average").

Thresholds and where they come from (cycles = time per launch x clock under that load; measured by this harness on
two boxes of round 5, profiles/r05_perf_guard.json, clock 2.38-2.40 GHz under every K3 variant):
  K3 config 2 (B = 8, 256x256, 9 scenes, tied roughness)   <= 93,000 cycles per launch      measured 85.5-87.1 k
  K3 config 2, MixedLoss (the training loss)               <= 98,500                        measured 91.2-91.4 k
  K3 config 2, untied roughness (three lobes)              <= 129,000                       measured 119.2-119.4 k
  K3 config-5 shape (B = 8, 512x512, 11 + 21 scenes)       <= 967,000                       measured 893.9-895.9 k
        = today + 7-8 %.  Round 3's kernel took 10 % longer than round 4's on one box (profiles/r04_k3_ab.txt: 39.7 vs
        35.9 us), i.e. ~96 k of these cycles at config 2: the bound sits between the two.
        NOTE on the 86 k the round-4 review proposed for config 2: that figure (and the 78-80 k "cycles per launch" of
        BENCH_r04.json / profiles/r04_bench.json) paired the timed region's duration with a clock read in a LATER
        interval by a probe of its own.  The clock under this kernel is not one number: it moves between 2.0 and 2.4 GHz
        within milliseconds, differs by box, and sags to ~1.75 GHz for ~5 ms when load follows an idle period
        (tools/clock_timeline.py, profiles/r05_clock_timeline*.txt) -- so that pairing is good to +-8 %, and it read low.
        Here the probe opens with the timed launches (same event) and spans ~90 % of them; by this harness today's build
        gives 85.5-88 k on boxes at 2.39 and at ~2.2 GHz.  bench.py pairs clock and duration of one interval since round 5.
  K1 / K2 (288 renders of 256x256, one per map) and K4 (64 samples) >= 0.72 of 8 TB/s
        BENCH_r04.json: 0.845 / 0.791 / 0.760; this harness: 0.840 / 0.808 / 0.779.  HBM-bound: judged in bytes per
        second (the HBM clock is not the shader clock).

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
K3_CONFIG2_MAX_CYCLES = 93_000
K3_MIXED_MAX_CYCLES = 98_500
K3_UNTIED_MAX_CYCLES = 129_000
K3_CONFIG5_MAX_CYCLES = 967_000
HBM_KERNELS_MIN_FRAC = 0.72

_fp = ctypes.c_void_p


def _maps(gen, B, H, tied=True):
    from bench import synthetic_maps
    return synthetic_maps(gen, B, H, tied=tied)


class Harness:
    def __init__(self, dev):
        from svbrdf_estimation_amd import _native
        self.native = _native
        self.lib = _native._load()
        self.dev = dev
        self.sa, self.sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        self.probe = self.lib.svbrdf_debug_clock_probe
        self.probe.argtypes = [_fp, ctypes.c_ulonglong, _fp]
        self.probe.restype = ctypes.c_int
        self.clk_out = torch.zeros(2, dtype=torch.int64, device=dev)
        assert self.probe(self.clk_out.data_ptr(), 1, _fp(self.sb.cuda_stream)) == 0     # first use loads the kernel (~6 ms of
        torch.cuda.synchronize(dev)                                                        # host time): not inside a region

    def measure(self, calls, est_us, want_clock=True, repeats=3):
        """calls: list of zero-argument callables, each enqueueing ONE launch on stream ``self.sa`` (visited round-robin).
        -> dict(us_per_launch, clock_GHz, cycles_per_launch), best repeat."""
        dev, sa, sb = self.dev, self.sa, self.sb
        n = int(max(12, min(120, 4000.0 / est_us)))          # ~4 ms of launches
        probe_ticks = int(max(20000, min(600000, 0.9 * n * est_us * 100)))   # ~90 % of the region, in 10 ns ticks
        # settle: clocks and caches in their steady state
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for k in range(32):
                calls[k % len(calls)]()
            torch.cuda.synchronize(dev)
        best = None
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            for k in range(n):                           # untimed prefill: the queue is deep before the region opens
                calls[k % len(calls)]()
            e0.record(sa)
            if want_clock:
                sb.wait_event(e0)
                assert self.probe(self.clk_out.data_ptr(), probe_ticks, _fp(sb.cuda_stream)) == 0
            t_host = time.perf_counter()
            for k in range(n):
                calls[(n + k) % len(calls)]()
            host_ms = 1e3 * (time.perf_counter() - t_host)
            e1.record(sa)
            torch.cuda.synchronize(dev)
            dev_ms = e0.elapsed_time(e1)
            # the host must stay ahead of the device, or the region holds idle gaps that are not the kernel's: with a
            # prefill as long as the region itself queued in front, it does as long as it enqueues the region's launches
            # in less time than the device needs to run them (measured: 0.55-0.6 of it for K3 at config 2, 0.02 for K1)
            assert host_ms < dev_ms, "host-bound: %.2f ms to enqueue %d launches the device ran in %.2f ms" % (host_ms, n, dev_ms)
            us = 1e3 * dev_ms / n
            ghz = None
            if want_clock:
                cyc, ticks = (int(v) for v in self.clk_out.tolist())
                ghz = cyc / ticks * 0.1
            res = {"us_per_launch": us, "launches": n, "clock_GHz": ghz,
                   "cycles_per_launch": us * ghz * 1e3 if ghz else None, "host_enqueue_ms": host_ms}
            if best is None or us < best["us_per_launch"]:
                best = res
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
        st = _fp(self.sa.cuda_stream)
        fn = lib.svbrdf_mixed_loss_fwd_bwd_host_scenes
        keep, calls = [table, xr, ws, loss], []
        for _ in range(sets):
            a, t = _maps(gen, B, H, tied).to(dev), _maps(gen, B, H, tied).to(dev)
            g = torch.empty_like(a)
            keep += [a, t, g]
            args = (a.data_ptr(), t.data_ptr(), table.data_ptr(), xr.data_ptr(), ctypes.c_float(0.1),
                    ctypes.c_float(l1_weight), ctypes.c_float(0.01), loss.data_ptr(), g.data_ptr(), ws.data_ptr(),
                    ws.numel() * 8, B, S, H, H, st)

            def call(args=args):
                rc = fn(*args)
                assert rc == 0, lib.svbrdf_last_error()
            calls.append(call)
        torch.cuda.synchronize(dev)
        return calls, keep, 144.0 * H * H * B

    def k12_calls(self, which):
        from svbrdf_estimation_amd import environment
        lib, dev = self.lib, self.dev
        B, H = 288, 256
        gen = torch.Generator().manual_seed(7)
        maps = _maps(gen, B, H).to(dev)
        torch.manual_seed(7)
        table = environment.BatchSceneSampler(B, 1, 0).sample().to(dev)
        xr = self.native.xrow(dev, H)
        st = _fp(self.sa.cuda_stream)
        if which == "k1":
            out = torch.empty(B, 1, 3, H, H, device=dev)
            args = (maps.data_ptr(), table.data_ptr(), xr.data_ptr(), out.data_ptr(), B, 1, H, H, st)
            fn, nbytes = lib.svbrdf_render_fwd, 60.0 * H * H * B
        else:
            cot = torch.randn(B, 1, 3, H, H, device=dev)
            out = torch.empty(B, 12, H, H, device=dev)
            args = (maps.data_ptr(), table.data_ptr(), xr.data_ptr(), cot.data_ptr(), out.data_ptr(), B, 1, H, H, st)
            fn, nbytes = lib.svbrdf_render_bwd, 108.0 * H * H * B
            maps = (maps, cot)

        def call():
            rc = fn(*args)
            assert rc == 0, lib.svbrdf_last_error()
        torch.cuda.synchronize(dev)
        return [call], [maps, table, xr, out], nbytes

    def k4_calls(self):
        lib, dev = self.lib, self.dev
        B, H = 64, 256
        gen = torch.Generator().manual_seed(17)
        lib.svbrdf_mix_materials.argtypes = [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]
        lib.svbrdf_mix_materials.restype = ctypes.c_int
        alpha = torch.rand(B, device=dev) * 0.8 + 0.1
        st = _fp(self.sa.cuda_stream)
        keep, calls = [alpha], []
        for _ in range(2):          # two alternating sets (604 MB), as in bench.py / tools/kernel_cases.py
            a, b = _maps(gen, B, H).to(dev), _maps(gen, B, H).to(dev)
            out = torch.empty_like(a)
            keep += [a, b, out]
            args = (a.data_ptr(), b.data_ptr(), alpha.data_ptr(), out.data_ptr(), B, H, H, st)

            def call(args=args):
                rc = lib.svbrdf_mix_materials(*args)
                assert rc == 0, lib.svbrdf_last_error()
            calls.append(call)
        torch.cuda.synchronize(dev)
        return calls, keep, 144.0 * H * H * B


@pytest.fixture(scope="module")
def harness():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (select CPU tests with -m 'not gpu')"
    return Harness(torch.device("cuda:0"))


_RESULTS = {}


def _record(name, res, nbytes):
    res = dict(res)
    res["algorithmic_bytes_per_launch"] = nbytes
    res["frac_of_hbm_peak"] = nbytes / (res["us_per_launch"] * 1e-6) / HBM_PEAK
    _RESULTS[name] = res
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "perf_guard.json"), "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "results": _RESULTS}, f, indent=1)
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


if __name__ == "__main__":
    h = Harness(torch.device("cuda:0"))
    for name, (mk, est, clk) in {
            "k3_config2": (lambda: h.k3_calls(8, 256, 3, 6), 38.0, True),
            "k3_config2_untied": (lambda: h.k3_calls(8, 256, 3, 6, tied=False), 52.0, True),
            "k3_config2_mixed": (lambda: h.k3_calls(8, 256, 3, 6, l1_weight=0.1), 42.0, True),
            "k3_config5_shape": (lambda: h.k3_calls(8, 512, 11, 21, sets=2), 400.0, True),
            "k1_288_renders": (lambda: h.k12_calls("k1"), 170.0, False),
            "k2_288_renders": (lambda: h.k12_calls("k2"), 325.0, False),
            "k4_64_samples": (lambda: h.k4_calls(), 100.0, False)}.items():
        if os.environ.get("PERF_GUARD_CASES") and name not in os.environ["PERF_GUARD_CASES"].split(","):
            continue
        calls, keep, nbytes = mk()
        for rep in range(int(os.environ.get("PERF_GUARD_REPEATS", "2"))):
            _record(name if rep == 0 else "%s_run%d" % (name, rep + 1), h.measure(calls, est, want_clock=clk), nbytes)
        del calls, keep
        torch.cuda.empty_cache()
    print(json.dumps(_RESULTS))
