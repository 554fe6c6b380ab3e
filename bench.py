#!/usr/bin/env python3
"""bench.py -- rendered 256x256 patches/s (fwd+bwd rendering loss) on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic input:
``loss = RenderingLoss(LocalRenderer())(input, target); loss.backward()`` -- scene
sampling on the host (reference RNG order), the [B,S,9] scene table handed to the launch by value,
the fused K3 kernel (both renderings, log/L1, analytic backward, in-kernel loss finalise) and the
gradient hand-over to ``input.grad``.  Inputs are resident in HBM before the timed region starts.

Workload = BASELINE.json configs[1]: synthetic 256x256 SVBRDF maps, 9 light/view samples
(3 random + 6 specular), per-GPU batch 8, fp32.  N > 1: one process per GPU (torchrun, or
self-spawned: ``python bench.py --gpus N``), batch sharded by rank, NO data-path collective (the
path is embarrassingly parallel); the only collectives are the barrier, the MAX over ranks of the
elapsed time and two small reporting reductions.

ONE CLOCK for the headline (round 6): `value`, `ms_per_step` and `roofline.achieved / frac` all come from the same
interval -- the wall time of the (median) timed region, MAX over ranks.  `value` is the TRAINING-LOOP figure: every step on
one stream, ``loss.backward()`` through PyTorch's autograd engine, which is what a network output gets (one kernel launch
per step: the engine, entered from the extension, is handed the cached device-resident 1.0 and the node skips its scale
launch).  Untimed follow-up legs of the same process report, beside it: the three-launch form of rounds 1-4 (the engine's
own ones-fill kernel and the node's scale launch), independent steps alternating on two streams, the shader clock under the loop (cycles per
launch), the copy bandwidth of this box (`copy_peak_GBps_measured`, svbrdf_debug_copy on 1 GiB) and the stand-alone
rates of the HBM-bound kernels against it.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(HBM-bound accounting of the dominant kernel k_rendering_loss: algorithmic bytes
144*H*W per patch / time per step of the timed region) and `cpu_baseline` (the eager-PyTorch
restatement of the reference's algorithm, oracle/eager_torch.py, timed on this box's
host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

T_PROCESS_START = time.perf_counter()

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

T_IMPORTED = time.perf_counter()

sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_secondary import HBM_PEAK_GBPS, copy_peak, replayed_counters, secondary_kernels  # noqa: E402,F401  (untimed legs)

FP32_VALU_PEAK_TFLOPS = 157.3   # ditto, packed-FMA vector peak
FLOP_PER_PIXEL_SCENE = 573.0    # SURVEY.md 8d (div/sqrt/log/pow counted as 1)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (config 2: 8)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--random-scenes", type=int, default=3)
    ap.add_argument("--specular-scenes", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--selftest", action="store_true",
                    help="first contact with a multi-GPU node (tools/scale_first_contact.md): after the bring-up, every rank "
                         "all-gathers its GPU's PCI address (N distinct devices expected), times an 8 MB and a 320 MB "
                         "all-reduce (DDP's payload) and checks the fused loss on its own device against a committed fixture of "
                         "the reference; prints one JSON line (ranks_seen, distinct_devices, allreduce_GBps, parity) and exits; "
                         "exit code 3 with a diagnosis on stderr when a check fails or anything hangs past --bringup-timeout")
    ap.add_argument("--selftest-expect-distinct", action="store_true", help=argparse.SUPPRESS)   # tests: make the self-test fail
    ap.add_argument("--timed-only", action="store_true",
                    help="settle, warm-up and the timed region(s) only: none of the untimed follow-up legs (clock probe, the other "
                         "issue pattern, the other backward modes, copy bandwidth).  For profiler passes: every launch of the "
                         "fused loss in the process is then a launch of the timed loop")
    ap.add_argument("--no-copy-peak", action="store_true", help="skip the 1 GiB copy-bandwidth leg (2 GiB of device memory)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the K1/K2 stand-alone rates (N=1 only)")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed device settle phase before the W warm-up steps: the same step function is run for "
                         "this long so that clocks (DVFS) and caches are in their steady state even when W is small; "
                         "reported in the JSON line")
    ap.add_argument("--backend", default=None, choices=("nccl", "gloo"),
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only for plumbing tests of the "
                         "multi-rank control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 (plumbing tests only, with --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still create the process group (world size 1) and take every multi-rank branch -- "
                         "RCCL init with device_id, barrier(device_ids=...), the MAX all-reduce of the elapsed time on a "
                         "device tensor, the global-mean all-reduce -- so that the code an N-GPU run executes can be "
                         "exercised on a one-GPU box (tests/test_gpu_multirank.py)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the steps are issued on, round-robin (step k runs entirely -- forward launch and "
                         "backward -- on stream k mod N).  Default 1: what a training loop, whose steps are serialised by the "
                         "optimizer, gets -- this is `value`.  Steps of a bench loop are independent batches, so with N = 2 "
                         "the ramp and tail of one step's kernel are filled by the next step's; that figure is reported "
                         "beside the headline as `two_streams_overlapped` (DESIGN.md section 5.1)")
    ap.add_argument("--rotate", type=int, default=6,
                    help="distinct (input, target) batches visited round-robin by the steps; 6 x 50 MB exceeds the "
                         "256 MiB Infinity Cache")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="multi-rank control flow only (rank launch, rendezvous, barrier, MAX over ranks, one JSON line "
                         "with ranks_seen) without touching a GPU: what the CPU test suite runs with --backend gloo")
    ap.add_argument("--regions", type=int, default=15,
                    help="short forms (--steps < 256): timed regions of --steps steps each, run back to back; `value` is the "
                         "median region (at least 9, default 15: a 20-step region is 0.74 ms and a host hiccup of a shared box "
                         "costs a whole one -- with N ranks the job's region time is the MAX over ranks, so one rank's hiccup "
                         "spoils it; the median of 15 tolerates seven; the 2000-step default form times one region)")
    ap.add_argument("--bringup-timeout", type=float, default=60.0,
                    help="N > 1: seconds the rendezvous + communicator set-up + first all-reduce may take before the rank "
                         "prints a diagnosis (backend, devices, HSA_ENABLE_IPC_MODE_LEGACY, ...) and exits with code 3.  The "
                         "limit covers the rendezvous: raise it for multi-node jobs and cold starts in which ranks page in their "
                         "images at very different speeds")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget of the main leg (the probes, the "
                    "one-thread figure and the C port add ~9 s)")
    ap.add_argument("--engine-threads", action="store_true",
                    help="keep PyTorch's multithreaded backward engine (default: run backward on the calling thread; "
                         "one process drives one GPU, the device-thread hop only adds wake-up latency)")
    return ap.parse_args()


def synthetic_maps(gen, B, H, rough_min=0.0, tied=True):
    """normals = normalize(0.3N, 0.3N, 1+|0.3N|); diffuse, specular ~ U(0,1);
    roughness ~ U(rough_min,1), one channel tiled x3 (model-like) unless tied=False."""
    n = torch.randn(B, 3, H, H, generator=gen) * 0.3
    n[:, 2] = 1.0 + n[:, 2].abs()
    n = n / n.norm(dim=1, keepdim=True)
    d = torch.rand(B, 3, H, H, generator=gen)
    r = torch.rand(B, 1 if tied else 3, H, H, generator=gen).expand(B, 3, H, H) * (1.0 - rough_min) + rough_min
    s = torch.rand(B, 3, H, H, generator=gen)
    return torch.cat((n, d, r, s), dim=1).contiguous()


def _time_eager(threads, inp, tgt, table, budget_s, max_patches=400):
    from oracle import eager_torch
    torch.set_num_threads(threads)
    xi = inp[:1].clone().requires_grad_(True)
    eager_torch.rendering_loss(xi, tgt[:1], table[:1]).backward()      # warm-up, untimed
    done, t0 = 0, time.perf_counter()
    while True:
        b = done % inp.shape[0]
        xi = inp[b:b + 1].clone().requires_grad_(True)
        eager_torch.rendering_loss(xi, tgt[b:b + 1], table[b:b + 1]).backward()
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= max_patches:
            return done / el, done, el


def _cpu_model():
    """first `model name` of /proc/cpuinfo ("core count and CPU model printed")"""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(args, inp, tgt, table):
    """Eager-PyTorch port of the reference's algorithm on the host cores (bounded sample, ~20 s in all).
    Eager 256x256 elementwise ops do not scale to hundreds of threads (at 256 threads one patch takes minutes:
    oversubscription), so a few thread counts are probed briefly and the fastest is the reported baseline
    (`cores` = threads used); every probed count is listed, the largest one also flat (`largest_probe_*`), and the
    all-cores figure BASELINE.md section 3 words is given only when the host is small enough for it to be one of them."""
    ncpu = os.cpu_count() or 1
    cands = sorted({min(ncpu, c) for c in (4, 8, 16, 32)})
    probe = {}
    for c in cands:
        probe[c] = _time_eager(c, inp, tgt, table, budget_s=1.0, max_patches=4)[0]
        print("[bench] cpu probe threads=%d: %.2f patches/s" % (c, probe[c]), file=sys.stderr, flush=True)
    best = max(probe, key=probe.get)
    rate, done, el = _time_eager(best, inp, tgt, table, budget_s=args.cpu_seconds)
    one, _, _ = _time_eager(1, inp, tgt, table, budget_s=2.0, max_patches=4)
    res = {"value": rate, "unit": "patches/s", "cores": best, "kind": "port",
           "cpu_model": _cpu_model(), "cpus": ncpu,
           # patches/s and renders/s (a patch = S scenes x (input + target) renders, forward + backward)
           "renders_per_s": rate * int(table.shape[1]) * 2,
           "probe_patches_per_s_by_threads": {str(c): probe[c] for c in cands},
           "largest_probe_threads": cands[-1], "largest_probe_patches_per_s": probe[cands[-1]],
           "sample": "%d patches of %dx%d, S=%d, fwd+bwd, eager PyTorch restatement (oracle/eager_torch.py), "
                     "%.1f s, best of threads %s on a %d-cpu host" % (done, args.size, args.size, table.shape[1], el,
                                                                     cands, ncpu),
           "one_thread_patches_per_s": one}
    if ncpu in probe:       # torch.set_num_threads(os.cpu_count()) as BASELINE.md section 3 states it: only where it was run
        res["all_cores_threads"], res["all_cores_patches_per_s"] = ncpu, probe[ncpu]
    try:   # the plain-C oracle (OpenMP), best of two thread counts, for orientation
        from oracle import c_oracle
        a, b, c = inp[:2].numpy(), tgt[:2].numpy(), table[:2].numpy()
        best_c = (0.0, 0)
        for th in sorted({min(ncpu, t) for t in (8, 64)}):
            c_oracle.set_threads(th)
            c_oracle.rendering_loss(a, b, c)
            t0, reps = time.perf_counter(), 0
            while time.perf_counter() - t0 < 1.0:
                c_oracle.rendering_loss(a, b, c)
                reps += 1
            best_c = max(best_c, (reps * 2 / (time.perf_counter() - t0), th))
        res["c_oracle_patches_per_s"], res["c_oracle_threads"] = best_c
    except Exception as e:  # pragma: no cover
        res["c_oracle_error"] = repr(e)
    return res


def median_region_index(job_elapsed):
    """index of the MEDIAN timed region (by the job's time = MAX over ranks of each region); with an even count the
    slower of the two middle regions, so that `value` never reads better than half of the regions did"""
    order = sorted(range(len(job_elapsed)), key=lambda k: job_elapsed[k])
    return order[len(order) // 2]


MEASURED_KEYS = (
    "B", "H", "S", "world", "n_batches", "elapsed", "job_elapsed", "median_region", "n_regions", "kernel_ms",
    "kernel_ms_avg", "region_ms_per_launch", "main_ns", "clock_ghz", "clock_note", "cycle_leg_ms",
    "other_ms_per_step", "other_ms", "other_steps", "leg_steps", "engine_ms_per_step",
    "engine_plain_ms_per_step", "copy", "mean_loss", "per_rank", "ranks_seen", "process_group", "host_path",
)


def assemble_line(args, m):
    """The JSON line from what main() measured (`m`: one value per MEASURED_KEYS).  Pure arithmetic and wording -- no GPU, no
    timing -- so the contract's identities (value x ms_per_step = patches per step; roofline.frac x peak x ms_per_step =
    algorithmic bytes; every follow-up figure None under --timed-only) are unit-tested on the CPU with made-up measurements
    (tests/test_bench_contract.py)."""
    from svbrdf_estimation_amd import _native
    (
        B, H, S, world, n_batches, elapsed, job_elapsed, median_region, n_regions, kernel_ms, kernel_ms_avg,
        region_ms_per_launch, main_ns, clock_ghz, clock_note, cycle_leg_ms, other_ms_per_step, other_ms,
        other_steps, leg_steps, engine_ms_per_step, engine_plain_ms_per_step, copy, mean_loss,
        per_rank, ranks_seen, process_group, host_path
    ) = (m[k] for k in MEASURED_KEYS)
    patches = world * B * args.steps
    ms_per_step = 1e3 * elapsed / args.steps      # THE interval of the headline: value, roofline.achieved and frac
    alg_bytes = 144.0 * H * H * B                 # per launch: 36 planes x 4 B per patch
    n_streams = max(1, main_ns)
    follow_up = "follow-up leg of rank 0, same process and tensors, right after the timed region"
    # the two ways of issuing the steps, whichever of them was the timed region
    if main_ns == 0:
        one = {"ms_per_step": ms_per_step, "kernel_ms": kernel_ms, "steps": args.steps, "leg": "the timed region"}
        two = {"ms_per_step": other_ms_per_step, "kernel_ms": other_ms, "steps": other_steps, "streams": 2, "leg": follow_up}
    else:
        one = {"ms_per_step": other_ms_per_step, "kernel_ms": other_ms, "steps": other_steps, "leg": follow_up}
        two = {"ms_per_step": ms_per_step, "kernel_ms": kernel_ms, "steps": args.steps, "streams": main_ns,
               "leg": "the timed region"}

    def rate(ms, n=1):          # patches/s of n GPUs at `ms` per step; None when the leg did not run (--timed-only)
        return n * B / (ms * 1e-3) if ms else None

    def avg(v):
        return sum(v) / len(v) if v else None
    one_kernel_avg, two_kernel_avg = avg(one["kernel_ms"]), avg(two["kernel_ms"])
    # ONE clock: the roofline is priced with the interval `value` is priced with (wall time of the median region, MAX
    # over ranks, / steps): frac x peak x ms_per_step / bytes == 1.  What rounds 1-5 priced it with -- the HIP event
    # pair around the region on the launch stream / launches, 2-3 % shorter (no barrier, no synchronize, no host tail) --
    # stays beside it as time_per_launch_ms / frac_by_launch_events.
    achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9
    events_ms = region_ms_per_launch if n_streams == 1 else None
    achieved_events = alg_bytes / (events_ms * 1e-3) / 1e9 if events_ms else None
    achieved_two = alg_bytes / (two["ms_per_step"] * 1e-3) / 1e9 if two["ms_per_step"] else None
    copy_gbps = copy.get("GBps") if copy else None
    valu_issue = None
    traffic, traffic_source, tj = replayed_counters(_native.library_path(), B, H, S)
    cycle_ms = cycle_leg_ms or events_ms or ms_per_step
    if tj is not None and tj.get("valu_wave_instr_per_launch") and clock_ghz:
        # what actually bounds K3: wave64 VALU instructions issued (PMC SQ_INSTS_VALU of the same
        # kernel, profiles/) against the SIMD-32 issue peak of one per 2 cycles per SIMD
        # (MI355X_MICROARCH.md), 1024 SIMDs, at the clock the chip holds under the loop
        peak = 1024 * clock_ghz * 1e9 / 2.0
        issue_rate = tj["valu_wave_instr_per_launch"] / (cycle_ms * 1e-3)      # duration of the clock's own interval
        valu_issue = {"wave_instr_per_launch": tj["valu_wave_instr_per_launch"],
                      "of_which_transcendental": tj.get("trans_wave_instr_per_launch"),
                      "instr_count_source": traffic_source,
                      "achieved_wave_instr_per_s": issue_rate, "peak_wave_instr_per_s": peak,
                      "frac": issue_rate / peak,
                      # a transcendental holds the SIMD for 6.5 plain issue slots when several waves share
                      # it (profiles/r01_valu_trans.txt: 4 rcp + 28 mul vs 32 mul, 4 waves per SIMD)
                      "frac_transcendental_weighted":
                          (issue_rate / peak) * (1.0 + 5.5 * tj["trans_wave_instr_per_launch"] / tj["valu_wave_instr_per_launch"])
                          if tj.get("trans_wave_instr_per_launch") else None,
                      "clock_GHz_under_load": clock_ghz, "clock_source": clock_note}
    working_set = n_batches * (2 * 12 + 12) * H * H * B * 4
    timed_is = ("loss.backward() through PyTorch's autograd engine (what a network output gets: the training-loop figure; one "
                "kernel launch per step)")
    out = {
        "metric": "rendered 256x256 patches/sec (fwd+bwd rendering loss)",
        "value": patches / elapsed, "unit": "patches/s", "n_gpus": world,
        "per_gpu_value": patches / elapsed / world,
        "value_is": timed_is,
        # the same figure under its round-5 name (then a follow-up leg; since round 6 it IS the timed region)
        "value_through_autograd_engine": patches / elapsed,
        "value_through_autograd_engine_with_fill_and_scale_launches": rate(engine_plain_ms_per_step, world),
        "valu_issue_frac": valu_issue["frac"] if valu_issue else None,      # what bounds K3 (also in roofline, with its inputs)
        "value_single_stream": rate(one["ms_per_step"], world),
        "value_two_streams_overlapped": rate(two["ms_per_step"], world),
        "follow_up_legs": "skipped (--timed-only)" if args.timed_only else "run",
        "copy_peak_GBps_measured": copy_gbps,
        "timed_regions": {"count": n_regions, "steps_each": args.steps, "median_index": median_region,
                          "ms_per_step": [1e3 * e / args.steps for e in job_elapsed],
                          "value": [world * B * args.steps / e for e in job_elapsed],
                          "spread_max_minus_min_over_median": (max(job_elapsed) - min(job_elapsed)) / elapsed,
                          "note": "each region: exactly `steps` steps between barrier + synchronize on both sides, MAX over "
                                  "ranks; `value`, `ms_per_step` and the roofline are the MEDIAN region's" if n_regions > 1 else
                                  "one region (the form with >= 256 steps)"},
        "value_note": (("`value` = the MEDIAN of %d consecutive timed regions of %d steps each (all listed in timed_regions); "
                        % (n_regions, args.steps)) if n_regions > 1 else "") +
                      "`value` = patches / wall time of the timed region: every step on %s, %s.  roofline.achieved and "
                      "roofline.frac are priced with the same interval (ms_per_step).  Untimed follow-up leg of the same "
                      "process: value_two_streams_overlapped = independent steps alternating on two streams (a bench-loop "
                      "property, not a training loop's)"
                      % ("ONE stream" if main_ns == 0 else "%d streams" % main_ns, timed_is),
        "ranks_seen": ranks_seen,       # summed by the bring-up all-reduce itself, not read from the environment
        "process_group": process_group,
        "per_rank": per_rank,
        "launch": "self-spawned" if os.environ.get("SVBRDF_SELF_SPAWNED") else
                  ("external launcher" if world > 1 else "single process"),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: synthetic %dx%d 12-channel SVBRDF maps, %d light/view "
                               "samples (%d random + %d specular), per-GPU batch %d, RenderingLoss fwd+bwd, %s"
                               % (H, H, S, args.random_scenes, args.specular_scenes, B,
                                  "one step at a time on one stream" if main_ns == 0 else
                                  "TWO INDEPENDENT BATCHES IN FLIGHT: steps alternate on %d streams" % main_ns),
                   "global_batch": world * B, "H": H, "W": H, "scenes": S,
                   "parallelism": "batch-sharded x%d, no data-path collective" % world,
                   "streams_per_gpu": n_streams,
                   "backward": "autograd engine",
                   "distinct_batches": n_batches,
                   "working_set_MiB": working_set / 2.0 ** 20,
                   "working_set_note": "input + target + gradient of every batch visited round-robin; the Infinity "
                                       "Cache holds 256 MiB"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "time_per_step_ms": ms_per_step,          # = ms_per_step: the interval achieved / frac are priced with
                     "achieved_definition": "algorithmic bytes per launch (144*H*W*B) / ms_per_step of the (median) timed "
                                            "region -- the interval `value` is priced with; one launch per step",
                     # secondary: the HIP event pair around the same region on the launch stream / launches (rounds 1-5's
                     # pricing; excludes the barrier / synchronize / host tail of the region: 2-3 % shorter)
                     "time_per_launch_ms": events_ms,
                     "frac_by_launch_events": achieved_events / HBM_PEAK_GBPS if achieved_events else None,
                     # against the copy bandwidth this box reached in this run (svbrdf_debug_copy, 1 GiB each way)
                     "copy_peak_GBps_measured": copy_gbps,
                     "frac_of_measured_copy_peak": achieved / copy_gbps if copy_gbps else None,
                     "copy_peak": copy,
                     # flat copies of what the nested objects below hold (a line parser that keeps scalars keeps these)
                     "valu_issue_frac": valu_issue["frac"] if valu_issue else None,
                     "valu_issue_clock_GHz": valu_issue["clock_GHz_under_load"] if valu_issue else None,
                     # boxes (and builds: the chip lowers its clock as the issue stream gets denser) differ in clock by
                     # 5-15 %; launch duration x the clock measured in this run is the box-independent figure
                     "shader_cycles_per_launch": (cycle_ms * 1e-3 * clock_ghz * 1e9) if clock_ghz else None,
                     "shader_cycles_leg_ms_per_launch": cycle_leg_ms,    # the interval the clock was read in (untimed follow-up leg)
                     "valu_issue_wave_instr_per_launch": valu_issue["wave_instr_per_launch"] if valu_issue else None,
                     "kernel": "%s<GRAD=true,L1=false,HEAD=false> (single launch: both shadings, log/L1, adjoint, loss finalise)"
                               % ("k_rendering_loss_inl" if B * S <= _native.host_scenes_max_rows() else "k_rendering_loss"),
                     "scene_table": "by value in the kernel-argument block (no H2D command)"
                                    if B * S <= _native.host_scenes_max_rows() else "pinned-ring upload",
                     "launches_in_flight": n_streams,
                     # event pairs around a SAMPLE of single launches (dispatch gap + kernel + event bubble each): evidence
                     # that a launch is what fills a step, never what the roofline is priced with
                     "kernel_launches_timed": len(kernel_ms),
                     "kernel_ms_avg": kernel_ms_avg, "kernel_ms_median": kernel_ms[len(kernel_ms) // 2] if kernel_ms else None,
                     "kernel_ms_note": "pairs bracket every 16th launch of an untimed 512-step leg right after the timed region(s) (no "
                                       "timed region carries timing events); a pair spans the ~3 us dispatch gap in front of "
                                       "the launch and its own end-of-pipe bubble, so it reads longer than ms_per_step",
                     "valu_frac_of_fp32_peak": (FLOP_PER_PIXEL_SCENE * H * H * S * B / (ms_per_step * 1e-3))
                                               / (FP32_VALU_PEAK_TFLOPS * 1e12),
                     "valu_issue": valu_issue},
        "single_stream": {"patches_per_s": rate(one["ms_per_step"]), "ms_per_step": one["ms_per_step"],
                          "kernel_ms_avg": one_kernel_avg,
                          "kernel_ms_median": one["kernel_ms"][len(one["kernel_ms"]) // 2] if one["kernel_ms"] else None,
                          "kernel_launches_timed": len(one["kernel_ms"]), "steps": one["steps"], "leg": one["leg"]},
        "two_streams_overlapped": {"patches_per_s": rate(two["ms_per_step"]), "ms_per_step": two["ms_per_step"],
                                   "streams": two["streams"], "kernel_ms_avg_while_overlapped": two_kernel_avg,
                                   "steps": two["steps"], "leg": two["leg"],
                                   "roofline_frac_of_time_share": achieved_two / HBM_PEAK_GBPS if achieved_two else None,
                                   "note": "per GPU: step k (launch + backward) on stream k mod 2; the steps are independent "
                                           "batches, so one step's kernel fills the ramp and tail of the other's.  A "
                                           "bench-loop property: not what one training loop gets"},
        "backward_modes": {
            "timed_region": "engine_one_launch_per_step",
            "engine_one_launch_per_step": {"patches_per_s": rate(engine_ms_per_step), "ms_per_step": engine_ms_per_step},
            "engine_with_fill_and_scale_launches": {"patches_per_s": rate(engine_plain_ms_per_step),
                                                    "ms_per_step": engine_plain_ms_per_step},
            "steps_each": leg_steps,
            "note": "per GPU, untimed follow-up legs on one stream, HIP events around each.  engine_one_launch_per_step (the timed "
                    "region's mode, once more): the engine, entered from the extension, is handed the cached device-resident 1.0 "
                    "and the node, recognising it by address and version, skips its scale launch; "
                    "..._with_fill_and_scale_launches: the engine's own ones-fill kernel and the node's no-op scale launch, as "
                    "in rounds 1-4 (three kernels).  The engine-free leaf shortcut of rounds 2-5 is gone (no faster any more)"},
        "loss": mean_loss,
        "host_path": host_path,
        "autograd_engine": "multithreaded" if args.engine_threads else "calling thread",
        "settle_ms": args.settle_ms,
    }
    return out, copy_gbps


def plumbing_only(args, rank, world, placement):
    """--plumbing-only: everything of the multi-rank protocol except the GPU work (CPU test of the launch path)."""
    import torch.distributed as dist
    seen = 1
    if world > 1:
        from svbrdf_estimation_amd import distributed
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        seen = distributed.init_process_group_checked("gloo", None, args.bringup_timeout)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    places = [placement]
    selftest = None
    if world > 1 and args.selftest:     # the collective part of the first-contact self-test, without a device (CPU suite)
        selftest = distributed.first_contact_selftest(None, False, False, None, args.bringup_timeout)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        places = [None] * seen
        dist.all_gather_object(places, placement)
    if rank == 0:
        print(json.dumps({"metric": "rendered 256x256 patches/sec (fwd+bwd rendering loss)", "value": None,
                          "unit": "patches/s", "n_gpus": world, "ranks_seen": seen, "plumbing_only": True,
                          "elapsed_max_over_ranks_s": float(t.item()),
                          "per_rank": {"cpus": [p["cpus"] for p in places], "numa_node": [p["numa_node"] for p in places],
                                       "cpu_binding": [p["source"] for p in places]},
                          "launch": "self-spawned" if os.environ.get("SVBRDF_SELF_SPAWNED") else "external launcher",
                          "selftest": selftest}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


class Rank:
    """what one rank process is and talks through: rank / world, its device, the process group (or None), its CPU placement"""

    def __init__(self, args):
        from svbrdf_estimation_amd import launch
        # this pool's host driver only supports dmabuf IPC: RCCL between rank processes needs it (read when the HSA runtime
        # starts, i.e. at the first GPU call; the self-spawning parent sets it for its children too)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if args.backend is None:        # RCCL (nccl) unless every rank shares cuda:0, which RCCL refuses
            args.backend = "gloo" if (args.share_device and self.world > 1) else "nccl"
        elif args.backend == "nccl" and args.share_device and self.world > 1:
            raise SystemExit("--share-device puts every rank on cuda:0, which RCCL does not support: use --backend gloo")
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (or let bench.py spawn them: run it "
                             "without a rank environment)" % (args.gpus, self.world))
        # first thing in a rank, before any GPU call: pin it to the CPUs next to its GPU (launch.py)
        self.host_cpus = os.sched_getaffinity(0)         # the CPU baseline leg gets the whole host back
        self.placement = launch.bind_rank_to_gpu_numa(self.local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", self.world)),
                                                      args.share_device)
        self.nccl = args.backend == "nccl"
        self.dist, self.dev, self.ranks_seen = None, None, 1

    def attach_device_and_group(self, args):
        """the GPU and, for N > 1 (or --force-dist), the process group: bring-up under the watchdog, one rank per GPU"""
        from svbrdf_estimation_amd import launch
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        # The rank's own CPU work is tiny tensors (the scene sampler).  With torch's default intra-op pool -- one thread per
        # core, 256 on the GPU box -- every host-side tensor op that does go parallel (generating the synthetic maps) leaves
        # the pool's workers spinning for ~100-200 ms afterwards, and the launch path of the main thread ran 3-4x slow for that
        # long (seen as a host-bound first leg after each new batch size; profiles/r03_dbg_mixed*.txt).  A small pool, and
        # time-based settling before every timed leg.  (The CPU baseline sets its own thread counts.)
        torch.set_num_threads(max(1, min(8, self.placement["n_cpus"])))
        if not args.share_device and torch.cuda.device_count() <= self.local_rank:
            raise SystemExit("local rank %d but only %d device(s) visible" % (self.local_rank, torch.cuda.device_count()))
        if args.share_device:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        # is this device the GPU the rank's CPUs were chosen for?  (PCI address from sysfs against the runtime's; undone if not)
        self.placement = launch.crosscheck_placement(self.placement, self.local_rank, self.host_cpus)
        if self.world > 1 or args.force_dist:
            import torch.distributed as dist
            from svbrdf_estimation_amd import distributed
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:                          # --force-dist without a launcher: a rendezvous of one
                os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            # rendezvous + communicator + ONE one-element all-reduce under a watchdog: a bring-up problem on a node this code
            # has never seen costs --bringup-timeout seconds and leaves a diagnosis on stderr (exit code 3), not the launcher's
            # limit.  nccl = RCCL on ROCm
            self.ranks_seen = distributed.init_process_group_checked("nccl" if self.nccl else "gloo", self.dev, args.bringup_timeout)
            self.dist = dist
            # every rank on a GPU of its own?  A scaling number measured with two ranks on one device is not a scaling
            # number: checked before anything is timed (exit code 3 with the reason)
            self.placement["runtime_pci"] = distributed.require_distinct_devices(self.dev, args.share_device, self.rank)[self.rank]

    def barrier(self):
        if self.dist is None:
            return
        if self.nccl:
            self.dist.barrier(device_ids=[self.local_rank])
        else:
            self.dist.barrier()

    def gather(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(out, obj)
        return out

    def launch_word(self):
        return "self-spawned" if os.environ.get("SVBRDF_SELF_SPAWNED") else ("external launcher" if self.world > 1 else "single process")


def run_selftest(args, R):
    """first contact with a multi-GPU node (tools/scale_first_contact.md): distinct devices, all-reduce rates, parity of the
    fused loss on every rank's device -- one JSON line, exit code 3 with a diagnosis when anything is off"""
    from svbrdf_estimation_amd import distributed
    if R.dist is None:
        raise SystemExit("--selftest checks a process group: use --gpus N with N > 1, or --gpus 1 --force-dist")
    res = distributed.first_contact_selftest(R.dev, R.nccl, args.share_device and not args.selftest_expect_distinct,
                                             os.path.join(ROOT, "tests", "golden", "g3_loss_7_s5.npz"), args.bringup_timeout)
    places = R.gather(R.placement)
    if R.rank == 0:
        res.update({"metric": "first-contact selftest (no throughput measured)", "selftest": True, "n_gpus": R.world,
                    "ranks_seen": R.ranks_seen, "process_group": "%s, world size %d" % (args.backend, R.dist.get_world_size()),
                    "launch": R.launch_word(),
                    "per_rank": {"cpus": [p["cpus"] for p in places], "numa_node": [p["numa_node"] for p in places],
                                 "pci_crosscheck": [p.get("pci_crosscheck") for p in places]}})
        print(json.dumps(res), flush=True)
    R.barrier()
    R.dist.destroy_process_group()


class StepLoop:
    """The bench loop: six rotating (input, target) batches resident in HBM, the loss module, and `step()` = one pass of the
    hot path -- ``inp.grad = None; loss = loss_fn(inp, tgt); loss.backward()`` -- issued on the current stream, or on stream
    k mod `ns` when `ns` streams are in use.  Also owns the optional per-launch event pairs (`run_sampled`): a timing event
    is an end-of-pipe timestamp -- each bracketed launch costs its stream ~10 us of idle -- so NO timed region carries them;
    they bracket every `stride`-th launch of untimed legs only and are evidence that one launch fills a step, never what
    anything is priced with."""

    def __init__(self, args, R):
        from svbrdf_estimation_amd import _hostext, _native, distributed, losses, renderers
        self.dev, self._native, self.losses = R.dev, _native, losses
        B, H = args.batch, args.size
        gen = torch.Generator().manual_seed(distributed.rank_seed(1234, R.rank))
        self.inp_h, self.tgt_h = synthetic_maps(gen, B, H), synthetic_maps(gen, B, H)       # also the CPU baseline's inputs
        self.batches = [(self.inp_h.to(R.dev).requires_grad_(True), self.tgt_h.to(R.dev))]
        for _ in range(1, max(1, args.rotate)):
            self.batches.append((synthetic_maps(gen, B, H).to(R.dev).requires_grad_(True), synthetic_maps(gen, B, H).to(R.dev)))
        self.streams = [torch.cuda.Stream(R.dev) for _ in range(max(2, args.streams))]
        self.loss_fn = losses.RenderingLoss(renderers.LocalRenderer())
        self.loss_fn.random_configuration_count = args.random_scenes
        self.loss_fn.specular_configuration_count = args.specular_scenes
        torch.manual_seed(distributed.rank_seed(313, R.rank))    # per-rank scene RNG
        self.ns = args.streams if args.streams > 1 else 0        # 0: every step on the current stream
        self.k = 0
        self.ext = _hostext.module()                             # native host path: event pairs are recorded inside the extension
        self._pair = None                                        # (begin, end) torch events of the NEXT launch, or None
        _native.set_launch_hook(None)
        if not args.engine_threads:
            torch.autograd.set_multithreading_enabled(False)
        # the two-stream legs visit the same leaf inputs from alternating streams on purpose; torch >= 2.9 warns about the
        # AccumulateGrad node's stream then (once per process, a paragraph on stderr)
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        # `value` is the training-loop figure: loss.backward() through PyTorch's autograd engine (what a network output gets;
        # one kernel launch per step: losses._FusedLossTensor)
        losses._UNIT_GRADIENT = True

    def _hook(self, phase):                                      # ctypes host path: called around the launch
        if self._pair is not None:
            self._pair[0 if phase == "begin" else 1].record(torch.cuda.current_stream(self.dev))

    def step(self):
        k = self.k
        self.k = k + 1
        if self.ns:
            torch.cuda.set_stream(self.streams[k % self.ns])     # the whole step (launch, backward) is issued on this stream
        inp, tgt = self.batches[k % len(self.batches)]
        inp.grad = None
        loss = self.loss_fn(inp, tgt)
        loss.backward()
        return loss

    def back_to_default_stream(self):
        if self.ns:
            torch.cuda.set_stream(torch.cuda.default_stream(self.dev))

    def settle_and_warm_up(self, settle_ms, warmup):
        torch.cuda.synchronize(self.dev)               # inputs were produced on the default stream
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < settle_ms:      # untimed, see --settle-ms
            for _ in range(64):
                self.step()
            torch.cuda.synchronize(self.dev)
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize(self.dev)

    def run_sampled(self, n, stride):
        """an UNTIMED leg of n steps with a HIP event pair around every stride-th launch -> (sorted ms per bracketed launch,
        wall ms per step of the leg)"""
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % stride == 0 else None
                 for i in range(n)]
        if self.ext is not None:
            for p in pairs:                       # create the raw hipEvent handles once
                if p is not None:
                    p[0].record()
                    p[1].record()
            torch.cuda.synchronize(self.dev)
        else:
            self._native.set_launch_hook(self._hook)
        t0 = time.perf_counter()
        for p in pairs:
            if p is not None:
                if self.ext is not None:
                    self.ext.set_timing_events(p[0].cuda_event, p[1].cuda_event)
                else:
                    self._pair = p
            self.step()
            self._pair = None
        torch.cuda.synchronize(self.dev)
        wall_ms = 1e3 * (time.perf_counter() - t0) / n
        self._native.set_launch_hook(None)
        self.back_to_default_stream()
        return sorted(p[0].elapsed_time(p[1]) for p in pairs if p is not None), wall_ms


def timed_regions(args, R, loop):
    """THE measurement.  One region = EXACTLY --steps steps between barrier + synchronize on both sides, its time the MAX over
    ranks.  The default form (2000 steps, 75 ms) times one.  A SHORT form (the driver's --steps 20 is 0.76 ms of GPU time)
    times `--regions` such regions back to back and the MEDIAN region is reported: a single 0.8 ms region swung by +-6 %
    between runs on one box in round 4 (197-224 k), SURVEY 8d asks for a median over >= 100 iterations, and a host hiccup of a
    shared box costs a whole region.  Every region is listed in the JSON line (`timed_regions`).
    -> n_regions, per-region local wall seconds, per-region HIP-event ms per launch (None with several streams), last losses"""
    n_regions = 1 if args.steps >= 256 else max(9, args.regions)
    dev = loop.dev
    local_elapsed, region_ms, lasts = [], [], []
    # one HIP event pair around each WHOLE region, on the stream the kernels are launched on (one stream only: with N streams
    # there is no single stream that sees every launch): a secondary figure (time_per_launch_ms), without the barrier /
    # synchronize / host tail of the region.  Created (a torch event is created at its first record) before anything is
    # timed: inside a 0.74 ms region two event creations would be 1 % of it.
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if not loop.ns else None
             for _ in range(n_regions)]
    for p in pairs:
        if p is not None:
            p[0].record(torch.cuda.current_stream(dev))
            p[1].record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    for region in pairs:
        R.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if region:
            region[0].record(torch.cuda.current_stream(dev))
        for _i in range(args.steps):
            last = loop.step()
        if region:
            region[1].record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        local_elapsed.append(time.perf_counter() - t0)   # this rank's K steps are done; the job's time is the MAX over ranks
        R.barrier()                         # closing bracket: every rank has finished before anything else happens
        torch.cuda.synchronize(dev)
        region_ms.append(region[0].elapsed_time(region[1]) / args.steps if region else None)
        lasts.append(last)
    loop.back_to_default_stream()
    return n_regions, local_elapsed, region_ms, lasts


def reduce_over_ranks(args, R, loop, local_elapsed, region_ms, lasts):
    """MAX over ranks of every region's time, the median region, and the per-rank record of it"""
    from svbrdf_estimation_amd import distributed
    B = args.batch
    job_elapsed = list(local_elapsed)
    if R.dist is not None:
        where = loop.dev if R.nccl else "cpu"
        t = torch.tensor(local_elapsed, dtype=torch.float64, device=where)
        R.dist.all_reduce(t, op=R.dist.ReduceOp.MAX)            # per region: the slowest rank's time
        job_elapsed = [float(v) for v in t.tolist()]
    median_region = median_region_index(job_elapsed)
    elapsed, last = job_elapsed[median_region], lasts[median_region]
    if R.dist is not None:
        mine = torch.tensor([local_elapsed[median_region], float(last.item()), float(distributed.rank_seed(313, R.rank))],
                            dtype=torch.float64, device=where)
        mean_loss = distributed.global_mean(last.detach() if R.nccl else last.detach().cpu()).item()
        every = [torch.zeros_like(mine) for _ in range(R.dist.get_world_size())]
        R.dist.all_gather(every, mine)            # for the record: each rank's own clock, last loss and scene seed
        places = R.gather(R.placement)
        per_rank = {"elapsed_s": [float(e[0]) for e in every],
                    "ms_per_step": [1e3 * float(e[0]) / args.steps for e in every],
                    "patches_per_s": [B * args.steps / float(e[0]) for e in every],
                    "last_loss": [float(e[1]) for e in every], "scene_seed": [int(e[2]) for e in every],
                    "cpus": [p["cpus"] for p in places], "numa_node": [p["numa_node"] for p in places],
                    "cpu_binding": [p["source"] for p in places], "pci_crosscheck": [p.get("pci_crosscheck") for p in places],
                    "runtime_pci": [p.get("runtime_pci") for p in places],
                    "distinct_devices": len({p.get("runtime_pci") for p in places})}
    else:
        mean_loss = last.item()
        per_rank = {"elapsed_s": [elapsed], "ms_per_step": [1e3 * elapsed / args.steps],
                    "patches_per_s": [B * args.steps / elapsed], "cpus": [R.placement["cpus"]],
                    "numa_node": [R.placement["numa_node"]], "cpu_binding": [R.placement["source"]],
                    "pci_crosscheck": [R.placement.get("pci_crosscheck")]}
    return job_elapsed, median_region, elapsed, region_ms[median_region], mean_loss, per_rank


def clock_leg(loop, ms_guess):
    """The clock the chip holds under this kernel is not one number: it moves between 2.0 and 2.4 GHz within milliseconds
    (power management), differs by box, and sags to ~1.75 GHz for ~5 ms when load arrives after an idle period
    (tools/clock_timeline.py, profiles/r05_clock_timeline*.txt).  Cycles per launch therefore need clock and duration from
    the SAME interval: an event A on the loop's stream opens the interval, the probe stream waits for A and then spins the
    one-wave probe for ~3 ms, and an event B closes the interval after as many steps as run in that time.  (Rounds 1-4
    paired the timed region's duration with a clock read in a later interval: good to +-8 %.)
    -> clock GHz | None, how it was measured, ms per launch of that very interval | None"""
    dev, native = loop.dev, loop._native
    clock_ghz, note, leg_ms = None, "not measured", None
    try:
        probe_stream = torch.cuda.Stream(dev)
        probe_out = torch.zeros(2, dtype=torch.int64, device=dev)
        native.clock_probe(probe_out, ticks=1, stream=probe_stream)     # first use loads the probe kernel (~6 ms of host time): not inside the interval
        torch.cuda.synchronize(dev)
        k_leg = int(max(16, min(4096, 3.2 / ms_guess)))        # steps that fill the probe's 3 ms (and a little more)
        for _ in range(max(64, k_leg, int(12.0 / ms_guess))):   # >= 12 ms of load first: past the onset sag, queue deep when A is recorded
            loop.step()
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        loop_stream = torch.cuda.current_stream(dev)
        ev_a.record(loop_stream)
        probe_stream.wait_event(ev_a)
        native.clock_probe(probe_out, ticks=300000, stream=probe_stream)     # 3 ms of the 100 MHz counter, from A on
        for _ in range(k_leg):
            loop.step()
        ev_b.record(loop_stream if not loop.ns else torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        cyc, ticks = (int(v) for v in probe_out.tolist())
        if ticks > 0:
            clock_ghz = cyc / ticks * 0.1
            leg_ms = ev_a.elapsed_time(ev_b) / k_leg if not loop.ns else None
            note = ("measured in this run: s_memtime / s_memrealtime of a one-wave probe kernel spinning %.1f ms on its own "
                    "stream beside %d steps of the bench loop, probe and steps opened by the same event (clock and duration "
                    "of one interval)" % (ticks * 1e-5, k_leg))
    except Exception as e:  # pragma: no cover
        note = "probe failed: %r" % (e,)
    loop.back_to_default_stream()
    return clock_ghz, note, leg_ms


def other_issue_pattern_leg(loop, steps):
    """the OTHER way of issuing the steps than the timed region's: it ran them on one stream -> now alternating on two (one
    step's kernel fills the ramp and tail of the other's: a bench-loop property), and vice versa"""
    main_ns = loop.ns
    loop.ns = 2 if main_ns == 0 else 0
    n = max(80, min(steps, 1024))
    stride = max(1, min(16 if loop.ns == 0 else 32, n // 5))
    for _ in range(32):                         # let the other issue pattern reach its steady state
        loop.step()
    torch.cuda.synchronize(loop.dev)
    ms, wall_ms = loop.run_sampled(n, stride)
    loop.ns = main_ns
    return wall_ms, ms, n


def backward_mode_leg(loop, unit, n):
    """an untimed leg on one stream with HIP events around it: the engine with the unit gradient (the timed region's own mode:
    one launch per step) or with its own ones-fill kernel and the node's no-op scale launch (rounds 1-4: three kernels)"""
    main_ns, loop.ns = loop.ns, 0
    loop.losses._UNIT_GRADIENT = unit
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 20e-3:       # past the clock sag that follows a synchronize
        for _ in range(32):
            loop.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(loop.dev))      # no synchronize in front: the loop keeps running into the timed steps
    for _ in range(n):
        loop.step()
    e1.record(torch.cuda.current_stream(loop.dev))
    torch.cuda.synchronize(loop.dev)
    loop.losses._UNIT_GRADIENT, loop.ns = True, main_ns
    return e0.elapsed_time(e1) / n


def main():
    args = parse_args()
    from svbrdf_estimation_amd import launch
    if args.gpus > 1 and not launch.launched_as_rank():
        # started as ONE plain process (`python bench.py --gpus N`): become the parent of N fresh rank processes.
        # Nothing has initialised the GPU runtime in this process and nothing will (see launch.py).
        os.environ["SVBRDF_SELF_SPAWNED"] = "1"
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    R = Rank(args)
    if args.plumbing_only:
        return plumbing_only(args, R.rank, R.world, R.placement)
    R.attach_device_and_group(args)
    if args.selftest:
        return run_selftest(args, R)

    loop = StepLoop(args, R)
    loop.settle_and_warm_up(args.settle_ms, args.warmup)
    n_regions, local_elapsed, region_ms, lasts = timed_regions(args, R, loop)
    job_elapsed, median_region, elapsed, region_ms_per_launch, mean_loss, per_rank = reduce_over_ranks(
        args, R, loop, local_elapsed, region_ms, lasts)

    # ---- untimed follow-up legs (same process, same tensors).  --timed-only skips them: the profiler passes, whose per-kernel
    # average should be the timed loop's launches and nothing else
    main_ns = loop.ns
    kernel_ms, clock_ghz, clock_note, cycle_leg_ms = [], None, "not measured (--timed-only)", None
    other_ms_per_step, other_ms, other_steps, leg_steps = None, [], 0, 0
    engine_ms_per_step = engine_plain_ms_per_step = copy = None
    if not args.timed_only:
        kernel_ms, _ = loop.run_sampled(512, 16)                 # event pairs around every 16th launch of 512 untimed steps
        clock_ghz, clock_note, cycle_leg_ms = clock_leg(loop, min(local_elapsed) * 1e3 / args.steps)
        other_ms_per_step, other_ms, other_steps = other_issue_pattern_leg(loop, args.steps)
        leg_steps = max(600, other_steps)
        engine_ms_per_step = backward_mode_leg(loop, True, leg_steps)
        engine_plain_ms_per_step = backward_mode_leg(loop, False, leg_steps)
        if R.rank == 0 and not args.no_copy_peak:               # the copy bandwidth of this box, measured in this run
            try:
                copy = copy_peak(loop.dev)
            except Exception as e:  # pragma: no cover
                copy = {"GBps": None, "error": repr(e)}

    if R.rank == 0:
        out, copy_gbps = assemble_line(args, {
            "B": args.batch, "H": args.size, "S": args.random_scenes + args.specular_scenes, "world": R.world,
            "n_batches": len(loop.batches), "elapsed": elapsed, "job_elapsed": job_elapsed,
            "median_region": median_region, "n_regions": n_regions, "kernel_ms": kernel_ms,
            "kernel_ms_avg": sum(kernel_ms) / len(kernel_ms) if kernel_ms else None,
            "region_ms_per_launch": region_ms_per_launch, "main_ns": main_ns, "clock_ghz": clock_ghz,
            "clock_note": clock_note, "cycle_leg_ms": cycle_leg_ms, "other_ms_per_step": other_ms_per_step, "other_ms": other_ms,
            "other_steps": other_steps, "leg_steps": leg_steps,
            "engine_ms_per_step": engine_ms_per_step, "engine_plain_ms_per_step": engine_plain_ms_per_step, "copy": copy,
            "mean_loss": mean_loss, "per_rank": per_rank, "ranks_seen": R.ranks_seen,
            "process_group": ("%s (%s), world size %d" % (args.backend, "RCCL" if R.nccl else "CPU transport, plumbing only",
                                                          R.dist.get_world_size())) if R.dist is not None else None,
            "host_path": "native C++ extension (csrc/host_ext.cpp)" if loop.ext is not None else "python + ctypes"})
        t_main = time.perf_counter()
        if R.world == 1 and not args.no_secondary:
            out["secondary"] = secondary_kernels(loop.dev, args.size, copy_gbps)
        t_secondary = time.perf_counter()
        if R.world == 1 and not args.no_cpu_baseline:
            os.sched_setaffinity(0, R.host_cpus)      # "the host cores of the box", not the GPU's socket only
            table = loop.loss_fn.sample_scene_table(args.batch)
            out["cpu_baseline"] = cpu_baseline(args, loop.inp_h, loop.tgt_h, table)
        else:
            out["cpu_baseline"] = None
        out["wall_s"] = {"imports_to_main": T_IMPORTED - T_PROCESS_START, "gpu_legs": t_main - T_IMPORTED,
                         "secondary": t_secondary - t_main, "cpu_baseline": time.perf_counter() - t_secondary}
        print(json.dumps(out), flush=True)
    if R.dist is not None:
        R.dist.destroy_process_group()


if __name__ == "__main__":
    main()
