#!/usr/bin/env python3
"""One kernel of the engine, launched repeatedly over a working set beyond the 256 MiB Infinity Cache -- the program
rocprofv3 wraps for the per-kernel evidence under profiles/ (tools/collect_profiles.sh): kernel-trace stats and, in
separate passes, the FETCH_SIZE / WRITE_SIZE counters.  Not product code.

    python3 tools/kernel_cases.py <case> [launches]
cases:  k1 k2 (288 renders, one per map)   photos (K1 + sensor noise + clamp: 288 photos, one per map)   copy (1 GiB -> 1 GiB)
        synthesis (synthesis.render_inputs(B = 8, one photo each), a call after the other: ONE kernel per call)
        k4 (64 samples)   k3_render k3_mixed k3_head k3_head_l1 k3_untied (config 2:
        B=8, 256x256, 9 scenes)   k3_config4 (B=16, mixed)   k3_config5 (B=8, 512x512, 11+21 scenes)
Prints one JSON line: case, kernel name pattern, algorithmic bytes per launch, launches.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import synthetic_maps  # noqa: E402
from svbrdf_estimation_amd import _native, environment  # noqa: E402


def main():
    case = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    H = 256
    if case in ("k1", "k2"):
        B = 288
        maps = synthetic_maps(gen, B, H).to(dev)
        torch.manual_seed(7)
        table = environment.BatchSceneSampler(B, 1, 0).sample().to(dev)
        cot = torch.randn(B, 1, 3, H, H, device=dev)
        if case == "k1":
            call, nbytes, kern = (lambda: _native.render_fwd(maps, table)), 60.0 * H * H * B, "k_render_fwd"
        else:
            call, nbytes, kern = (lambda: _native.render_bwd(maps, table, cot)), 108.0 * H * H * B, "k_render_bwd"
    elif case == "photos":
        from svbrdf_estimation_amd import synthesis
        B = 288
        maps = synthetic_maps(gen, B, H).to(dev)
        torch.manual_seed(7)
        table = torch.stack([synthesis.input_scene_table(1, True) for _ in range(B)]).to(dev)
        levels = synthesis.noise_levels(B).view(B, 1).to(dev)
        state = {"k": 0}

        def call():
            state["k"] += 1
            _native.render_inputs(maps, table, levels, 99, 4 * state["k"])
        nbytes, kern = 60.0 * H * H * B, "k_render_inputs"
    elif case == "copy":
        nf = (1 << 30) // 4
        src = torch.empty(nf, device=dev).uniform_(-1.0, 1.0)
        dst = torch.empty_like(src)
        call, nbytes, kern = (lambda: _native.debug_copy(dst, src)), 8.0 * nf, "k_copy_vec4"
    elif case == "synthesis":
        from svbrdf_estimation_amd import synthesis
        B = 8
        maps = synthetic_maps(gen, B, H).to(dev)
        torch.manual_seed(7)
        call, nbytes, kern = (lambda: synthesis.render_inputs(maps, 1)), 60.0 * H * H * B, "k_render_inputs"
    elif case == "k4":
        B = 64
        sets = [(synthetic_maps(gen, B, H).to(dev), synthetic_maps(gen, B, H).to(dev)) for _ in range(2)]
        alpha = torch.rand(B, device=dev) * 0.8 + 0.1
        state = {"k": 0}

        def call():
            a, b = sets[state["k"] % 2]
            state["k"] += 1
            _native.mix_materials(a, b, alpha)
        nbytes, kern = 144.0 * H * H * B, "k_mix_materials"
    else:
        B, n_random, n_specular, kw, tied = 8, 3, 6, {}, True
        head = case in ("k3_head", "k3_head_l1")
        if case in ("k3_mixed", "k3_head_l1", "k3_config4"):
            kw["l1_weight"] = 0.1
        if case == "k3_untied":
            tied = False
        if case == "k3_config4":
            B = 16
        if case == "k3_config5":
            H, n_random, n_specular = 512, 11, 21
        rot = 6 if H == 256 else 2
        sets = []
        for _ in range(rot):
            a, t = synthetic_maps(gen, B, H, tied=tied).to(dev), synthetic_maps(gen, B, H, tied=tied).to(dev)
            if head:
                a = (torch.rand(B, 9, H, H, generator=gen) * 1.8 - 0.9).to(dev)        # the generator's tanh output
            sets.append((a, t))
        torch.manual_seed(11)
        tab = environment.BatchSceneSampler(B, n_random, n_specular).sample()
        state = {"k": 0}

        def call():
            a, t = sets[state["k"] % rot]
            state["k"] += 1
            _native.rendering_loss(a, t, tab, 0.1, want_grad=True, head=head, **kw)
        nbytes = (120.0 if head else 144.0) * H * H * B
        kern = "k_rendering_loss"
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    print(json.dumps({"case": case, "kernel": kern, "algorithmic_bytes_per_launch": nbytes, "launches": n + 5}))


if __name__ == "__main__":
    main()
