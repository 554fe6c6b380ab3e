"""Untimed legs of bench.py (SURVEY 8d's secondary figures), kept out of the file that holds the timed region:

  * ``copy_peak``           the copy bandwidth of this box, measured in the run (svbrdf_debug_copy on 1 GiB each way)
  * ``secondary_kernels``   K1 / K2 / K4 / the noise-fused K1 alone against 8 TB/s and against that copy bandwidth; renders/s
                            through the plugin interface (device and host tensors); input synthesis per call; K3 at the other
                            BASELINE shapes (configs[3], untied roughness, config 5's shape)
  * ``replayed_counters``   HBM bytes / VALU instructions of the headline kernel from profiles/k3_hbm_traffic.json, replayed
                            only for the very machine code they were recorded from

None of this is inside a timed region; nothing here imports the CPU port (bench.py's cpu_baseline leg is the only place)."""
import json
import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def replayed_counters(library, B, H, S, record_path=None):
    """Hardware-counter figures of the headline kernel (HBM bytes, VALU instructions per launch) are NOT measured in a
    bench run -- rocprofv3 --pmc needs its own passes (tools/collect_profiles.sh) -- but replayed from
    profiles/k3_hbm_traffic.json.  A replay is honest only for the very code the counters were taken from: the record
    carries the sha256 of the kernel's instruction bytes (svbrdf_estimation_amd/_codehash.py) and is replayed only when the
    kernel inside `library` hashes to it and the shape is the recorded one.
    -> (hbm bytes per launch | None, what was done and why (str) | None, the record (dict) when replayed else None)"""
    import hashlib
    path = record_path or os.path.join(ROOT, "profiles", "k3_hbm_traffic.json")
    if not os.path.exists(path):
        return None, None, None
    try:
        with open(path, "rb") as f:
            raw = f.read()
        tj = json.loads(raw.decode())
        if not (tj.get("B") == B and tj.get("H") == H and tj.get("S") == S):
            return None, "NOT replayed: %s holds the shape B=%s H=%s S=%s" % (os.path.basename(path), tj.get("B"), tj.get("H"), tj.get("S")), None
        from svbrdf_estimation_amd import _codehash
        try:
            have = _codehash.k3_headline_hash(library)["sha256"]
        except Exception as e:
            have = "unreadable (%r)" % (e,)
        if not tj.get("kernel_code_sha256") or tj["kernel_code_sha256"] != have:
            return None, ("NOT replayed: profiles/k3_hbm_traffic.json holds counters of kernel code sha256 %s, the kernel in %s "
                          "is %s -- re-record them (tools/collect_profiles.sh + summarize_profiles.py)"
                          % (str(tj.get("kernel_code_sha256"))[:16], os.path.basename(library), have[:16])), None
        return (tj.get("hbm_bytes_per_launch"),
                "NOT measured in this run: PMC counters of the same kernel and shape recorded with rocprofv3 --pmc by "
                "tools/collect_profiles.sh, replayed from profiles/k3_hbm_traffic.json (sha1 %s, build %s, kernel code sha256 "
                "%s = this library's)" % (hashlib.sha1(raw).hexdigest()[:12], tj.get("git_head", "?"), have[:16]), tj)
    except Exception as e:
        return None, "NOT replayed: %r" % (e,), None


def synthetic_maps_on_device(dev, seed, B, H, rough_min=0.0, tied=True):
    """synthetic_maps' distribution, drawn by the device generator: for the untimed secondary legs, whose 288-map working
    sets take seconds to draw on the host (the headline inputs stay host-drawn and seeded per rank)"""
    gen = torch.Generator(device=dev).manual_seed(seed)
    n = torch.randn(B, 3, H, H, generator=gen, device=dev) * 0.3
    n[:, 2] = 1.0 + n[:, 2].abs()
    n = n / n.norm(dim=1, keepdim=True)
    d = torch.rand(B, 3, H, H, generator=gen, device=dev)
    r = torch.rand(B, 1 if tied else 3, H, H, generator=gen, device=dev).expand(B, 3, H, H) * (1.0 - rough_min) + rough_min
    s = torch.rand(B, 3, H, H, generator=gen, device=dev)
    return torch.cat((n, d, r, s), dim=1).contiguous()


def _event_timed(fn, reps, dev, warm=3):
    """average ms per call of `fn` over `reps` back-to-back calls, HIP events on the current stream (the launch stream)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / reps


def copy_peak(dev, gib=1.0):
    """"report fraction of both nominal and measured-copy peak" -- the copy bandwidth of THIS box, measured in
    this run: svbrdf_debug_copy (float4 streaming copy, non-temporal) on `gib` GiB -> `gib` GiB, far beyond the 256 MiB
    Infinity Cache; bytes moved = read + written.  Best of three 10-launch regions."""
    from svbrdf_estimation_amd import _native
    n = int(gib * 2 ** 30) // 4
    src = torch.empty(n, device=dev).uniform_(-1.0, 1.0)
    dst = torch.empty_like(src)
    ms = min(_event_timed(lambda: _native.debug_copy(dst, src), 10, dev) for _ in range(3))
    ok = bool(torch.equal(src, dst))
    del src, dst
    return {"GBps": 8.0 * n / (ms * 1e-3) / 1e9, "ms_per_launch": ms, "bytes_moved_per_launch": 8.0 * n, "copied_correctly": ok,
            "kernel": "svbrdf_debug_copy: k_copy_vec4<1, nontemporal>, %.0f MiB read + %.0f MiB written per launch"
                      % (4.0 * n / 2 ** 20, 4.0 * n / 2 ** 20)}


def secondary_kernels(dev, H, copy_gbps):
    """K1 / K2 alone at one render per map with a working set far beyond the 256 MiB Infinity Cache
    (288 renders: 1.1 GB / 2.0 GB per launch): the HBM-bound kernels of the engine, for the record, against the nominal
    8 TB/s and against the copy bandwidth measured in this run (`copy_gbps`)."""
    from svbrdf_estimation_amd import _native, environment

    def hbm(gbps):
        return {"algorithmic_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
                "frac_of_measured_copy_peak": gbps / copy_gbps if copy_gbps else None}
    B = 288
    maps = synthetic_maps_on_device(dev, 7, B, H)
    torch.manual_seed(7)
    table = environment.BatchSceneSampler(B, 1, 0).sample().to(dev)
    cot = torch.randn(B, 1, 3, H, H, device=dev)
    out = {}
    for name, fn, nbytes in (("K1_render_fwd", lambda: _native.render_fwd(maps, table), 60.0 * H * H * B),
                             ("K2_render_bwd", lambda: _native.render_bwd(maps, table, cot), 108.0 * H * H * B)):
        ms = _event_timed(fn, 10, dev)
        out[name] = dict({"renders_per_launch": B, "ms_per_launch": ms, "renders_per_s": B / (ms * 1e-3)},
                         **hbm(nbytes / (ms * 1e-3) / 1e9))
    # K1 with the sensor-noise epilogue (svbrdf_render_inputs: + sigma * N(0,1), clamp): one photo per map, same bytes
    levels = torch.full((B, 1), 0.005, device=dev)
    ms = _event_timed(lambda: _native.render_inputs(maps, table, levels, 1, 4), 10, dev)
    out["K1_render_inputs_noise_clamp"] = dict({"photos_per_launch": B, "ms_per_launch": ms, "photos_per_s": B / (ms * 1e-3)},
                                               **hbm(60.0 * H * H * B / (ms * 1e-3) / 1e9))
    del maps, cot, levels
    # K3 alone (kernel-limited rates): sensitivity to the roughness distribution at config 2, the
    # three-lobe path (independent roughness channels), and config 5 (512x512, 11 + 21 scenes)
    def k3(tag, B, Hk, n_random, n_specular, **kw):
        a, t = synthetic_maps_on_device(dev, 11, B, Hk, **kw), synthetic_maps_on_device(dev, 12, B, Hk, **kw)
        torch.manual_seed(11)
        tab = environment.BatchSceneSampler(B, n_random, n_specular).sample()
        tab = tab if B * (n_random + n_specular) <= _native.host_scenes_max_rows() else tab.to(dev)
        call = lambda: _native.rendering_loss(a, t, tab, 0.1, want_grad=True)
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < 0.2:
            for _ in range(20):
                call()
            torch.cuda.synchronize(dev)
        # the ctypes binding costs ~50 us of host time per call, more than the kernel at config 2: events around EVERY
        # launch give the kernel's own duration (median), the wall time of the loop the call rate of this binding
        pairs = []
        t0 = time.perf_counter()
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize(dev)
        wall_ms = 1e3 * (time.perf_counter() - t0) / 30
        ms = sorted(p[0].elapsed_time(p[1]) for p in pairs)[len(pairs) // 2]
        gb = 144.0 * Hk * Hk * B / (ms * 1e-3) / 1e9
        out[tag] = {"B": B, "H": Hk, "scenes": n_random + n_specular, "ms_per_launch": ms,
                    "ms_per_call_wall_ctypes_binding": wall_ms,
                    "patches_per_s": B / (ms * 1e-3), "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBPS}
    def k3_module(tag, B, Hk, loss_fn, n_streams):
        """whole steps through the module interface (host path, autograd), like the headline loop"""
        sets = [(synthetic_maps_on_device(dev, 13 + 2 * q, B, Hk).requires_grad_(True), synthetic_maps_on_device(dev, 14 + 2 * q, B, Hk))
                for q in range(4)]
        sts = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else None
        torch.cuda.synchronize(dev)

        def run(n):
            for k in range(n):
                if sts:
                    torch.cuda.set_stream(sts[k % n_streams])
                a, t = sets[k % 4]
                a.grad = None
                loss_fn(a, t).backward()
        # settle by TIME, not by step count: right after the host-side generation of a new batch size the first ~100 ms of a
        # leg have been seen running 3-4x slow (host-bound: the intra-op pool's workers still spinning, see main())
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < 0.3:
            run(60)
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(300)
        torch.cuda.synchronize(dev)
        ms = 1e3 * (time.perf_counter() - t0) / 300
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
        gb = 144.0 * Hk * Hk * B / (ms * 1e-3) / 1e9
        out[tag] = {"B": B, "H": Hk, "streams": n_streams, "ms_per_step": ms, "patches_per_s": B / (ms * 1e-3),
                    "algorithmic_GBps": gb, "frac_of_hbm_peak": gb / HBM_PEAK_GBPS}
    from svbrdf_estimation_amd import losses, renderers, synthesis

    # SURVEY 8d's secondary metric: renders/s THROUGH the plugin interface, `LocalRenderer().render(scene, svbrdf)`
    # (renderers.py:67-104), one reference-shaped call after the other: a Scene object of python lists, one [12,H,W]
    # map (-> [1,3,H,W]) or a [8,12,H,W] batch with the one scene; forward, and forward + backward of a cotangent.
    # The call is one dispatch (the scene's nine floats ride in the launch's argument block), so with a 256x256 map it is
    # host-bound: us_per_call is host time.
    R = renderers.LocalRenderer()
    scene = environment.Scene(environment.Camera([0.1, -0.2, 2.0]), environment.Light([0.4, 0.3, 1.5], [30.0, 30.0, 30.0]))
    for tag, nb in (("LocalRenderer_render_one_map", 0), ("LocalRenderer_render_batch8", 8)):
        m = synthetic_maps_on_device(dev, 17 + nb, max(nb, 1), H)
        m = m if nb else m[0]
        cot = torch.randn(max(nb, 1), 3, H, H, device=dev)
        x = m.clone().requires_grad_(True)
        res = {}
        for mode in ("fwd", "fwd_bwd"):
            def call():
                if mode == "fwd":
                    R.render(scene, m)
                else:
                    x.grad = None
                    R.render(scene, x).backward(cot)
            t_settle = time.perf_counter()          # settle by time (see k3_module below)
            while time.perf_counter() - t_settle < 0.3:
                for _ in range(50):
                    call()
                torch.cuda.synchronize(dev)
            n = 500
            t0 = time.perf_counter()
            for _ in range(n):
                call()
            host_s = time.perf_counter() - t0        # the host's share: every call issued, the GPU still working
            torch.cuda.synchronize(dev)
            wall = time.perf_counter() - t0
            res[mode] = {"renders_per_s": n * max(nb, 1) / wall, "us_per_call": 1e6 * wall / n,
                         "host_us_per_call": 1e6 * host_s / n}
        res["maps_per_call"] = max(nb, 1)
        out[tag] = res
    # the dataloader's call shape served on the GPU (round 6): a HOST [1,12,H,W] map in, a HOST photo out (dataset.py:206-212:
    # pinned round trip around K1), PCIe-inclusive by construction
    host_map = synthetic_maps_on_device(dev, 19, 1, H).cpu()
    for _ in range(5):
        R.render(scene, host_map)
    t0, n = time.perf_counter(), 200
    for _ in range(n):
        R.render(scene, host_map)
    dt = (time.perf_counter() - t0) / n
    out["LocalRenderer_render_host_tensor"] = {
        "renders_per_s": 1.0 / dt, "us_per_call": 1e6 * dt,
        "note": "CPU tensor in, CPU tensor out (the reference dataloader's call): host copy into pinned memory, H2D, K1, D2H, "
                "event wait, clone -- 3.0 MiB up and 0.75 MiB down over PCIe per call"}
    # K4 (material mixing, dataset.py:142-160) and the input-photo synthesis on K1 (dataset.py:162-221), per call
    Bm = 64
    a, b = synthetic_maps_on_device(dev, 21, Bm, H), synthetic_maps_on_device(dev, 22, Bm, H)
    alpha = torch.rand(Bm, device=dev) * 0.8 + 0.1
    ms = _event_timed(lambda: _native.mix_materials(a, b, alpha), 20, dev)
    out["K4_mix_materials"] = dict({"samples_per_launch": Bm, "ms_per_launch": ms, "samples_per_s": Bm / (ms * 1e-3),
                                    "working_set_MiB": 36.0 * H * H * 4 * Bm / 2 ** 20}, **hbm(144.0 * H * H * Bm / (ms * 1e-3) / 1e9))
    for views, Bs in ((1, 8), (5, 16)):
        sv = a[:Bs]
        for _ in range(3):
            synthesis.render_inputs(sv, views)
        torch.cuda.synchronize(dev)
        launches = _native.launch_count()
        t0, n = time.perf_counter(), 30
        for _ in range(n):
            synthesis.render_inputs(sv, views)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / n
        out["render_inputs_B%d_views%d" % (Bs, views)] = {
            "photos_per_s": Bs * views / dt, "ms_per_call": 1e3 * dt,
            "kernel_launches_per_call": (_native.launch_count() - launches) / n,
            "note": "scene draws in the extension (the reference's draws in the reference's order) + ONE launch of K1 with the "
                    "noise + clamp epilogue (svbrdf_render_inputs_host_scenes), whole batch"}
    del a, b
    mixed = losses.MixedLoss(renderers.LocalRenderer())
    # BASELINE configs[3]: batch 16, mixed loss (the multi-view network's output has the same loss shapes); its 144 scene
    # rows ride in the launch's argument block like config 2's 72
    k3_module("K3_config4_B16_mixed_loss", 16, H, mixed, 1)
    k3_module("K3_config4_B16_mixed_loss_2streams", 16, H, mixed, 2)
    k3("K3_config2_roughness_U(0.2,1)", 8, H, 3, 6, rough_min=0.2)
    k3("K3_config2_untied_roughness", 8, H, 3, 6, tied=False)
    k3("K3_config5_512_32scenes", 8, 512, 11, 21)
    return out
