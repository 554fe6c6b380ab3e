"""The bench line's contract (task statement, section 4), checked on the lines committed under profiles/ -- bench.py itself
needs a GPU -- and, for the keys, on bench.py's source: one JSON object with the driver's fields, `roofline` and
`cpu_baseline` objects whose SCALAR keys carry what the driver's parser keeps, values mutually consistent."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = ("r05_bench.json", "r05_bench_short.json", "r04_bench.json", "r04_bench_short.json")


@pytest.mark.parametrize("name", LINES)
def test_committed_bench_line_meets_the_contract(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        b = json.load(f)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert b["metric"].startswith("rendered 256x256 patches/sec") and "patches/sec" in base["metric"]
    assert b["unit"] == "patches/s" and b["n_gpus"] == 1 and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["vs_baseline"] is None and base["published"] == {}             # no published number for this metric
    assert b["dtype"] == "f32" and b["data"] == "synthetic" and "configs[1]" in b["config"]["workload"]
    assert "model" not in b["config"] and b["config"]["global_batch"] == 8 and b["config"]["scenes"] == 9
    assert b["steps"] > 0 and b["warmup"] >= 0 and b["ms_per_step"] > 0
    # value = patches / timed region; per-GPU value and the through-the-engine figure sit next to it
    assert abs(b["value"] - 8 * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"] and b["per_gpu_value"] == b["value"]
    assert 0 < b["value_through_autograd_engine"] <= b["value"] * 1.05
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    alg = 144.0 * 256 * 256 * 8                                              # 36 planes x 4 B per patch
    assert r["algorithmic_bytes_per_launch"] == alg
    assert abs(r["achieved"] - alg / (r["time_per_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["time_per_launch_ms"] <= b["ms_per_step"] * 1.0001 and r["consistent_time_per_launch_le_ms_per_step"] is True
    assert r["kernel_launches_timed"] >= 32                                  # per-launch samples, also in the 20-step form
    assert alg <= r["traffic"] <= 1.05 * alg                                 # PMC bytes: no wasted re-reads
    # what bounds this kernel, as scalars: VALU issue fraction, the clock it was priced at, cycles per launch
    assert 0.3 < r["valu_issue_frac"] < 1.0 and 1.0 < r["valu_issue_clock_GHz"] < 2.6
    # cycles = duration x clock of ONE interval: since round 5 an untimed leg opened by the same event as its clock probe
    leg_ms = r.get("shader_cycles_leg_ms_per_launch") or r["time_per_launch_ms"]
    assert abs(r["shader_cycles_per_launch"] - leg_ms * 1e-3 * r["valu_issue_clock_GHz"] * 1e9) < 1.0
    if name.startswith("r05"):
        # short forms time nine regions and report the median; the replayed counters are keyed to the kernel's code
        tr = b["timed_regions"]
        assert tr["count"] == (9 if b["steps"] < 256 else 1) and len(tr["value"]) == tr["count"]
        assert abs(sorted(tr["value"])[tr["count"] // 2] - b["value"]) <= 1e-9 * b["value"]
        assert "kernel code sha256" in r["traffic_source"] and b["per_rank"]["pci_crosscheck"] == ["match"]
        assert 70e3 < r["shader_cycles_per_launch"] < 93e3                  # the GPU suite's guard, on the bench line
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "patches/s" and c["value"] > 0 and 1 <= c["cores"] <= c["all_cores_threads"]
    assert "patches" in c["sample"] and ("all_cores_patches_per_s" in c)
    assert len(b["per_rank"]["cpus"]) == 1 and isinstance(b["per_rank"]["cpus"][0], str)
    assert b["value"] / c["value"] > 1e3                                     # a reported baseline, not a target


@pytest.mark.parametrize("name", ("r06_bench.json", "r06_bench_short.json", "r06_bench_short2.json"))
def test_round6_bench_line_one_clock_copy_peak_cpu_fields(name):
    """VERDICT round 5, items 2-4, on the committed round-6 lines: `value` is the training-loop figure and `roofline.frac` is
    priced with value's own interval (frac x 8e12 x ms_per_step / bytes = 1.000); the copy bandwidth is measured in the run
    and the HBM-bound kernels are priced against it; the CPU baseline states model and core count, every probed thread count
    is a number, and the run is short."""
    with open(os.path.join(ROOT, "profiles", name)) as f:
        b = json.load(f)
    assert b["metric"].startswith("rendered 256x256 patches/sec") and b["unit"] == "patches/s" and b["n_gpus"] == 1
    assert b["dtype"] == "f32" and b["data"] == "synthetic" and "configs[1]" in b["config"]["workload"] and "model" not in b["config"]
    assert b["vs_baseline"] is None and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert "autograd engine" in b["value_is"] and b["config"]["backward"] == "autograd engine"
    assert b["value"] == b["value_through_autograd_engine"] == b["per_gpu_value"] and "value_leaf_shortcut" not in b
    assert abs(b["value"] - 8 * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"]
    r = b["roofline"]
    alg = 144.0 * 256 * 256 * 8
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["algorithmic_bytes_per_launch"] == alg
    assert abs(r["frac"] * 8e12 * b["ms_per_step"] * 1e-3 / alg - 1.0) < 1e-9          # ONE clock
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["time_per_step_ms"] == b["ms_per_step"]
    assert r["time_per_launch_ms"] <= b["ms_per_step"] * 1.0001 and r["frac_by_launch_events"] >= r["frac"] * 0.9999
    assert alg <= r["traffic"] <= 1.05 * alg and "kernel code sha256" in r["traffic_source"]
    # (the bench's cycle leg is ONE ~3 ms interval through the whole host path, not the guard's best of three from C: a
    # disturbed moment of a shared box shows in it -- 90.05 k in r06_bench_short2.json -- so the bound is the guard's + 3 %)
    assert 70e3 < r["shader_cycles_per_launch"] < 93e3 and 0.3 < r["valu_issue_frac"] < 1.0
    tr = b["timed_regions"]
    assert (tr["count"] >= 9 if b["steps"] < 256 else tr["count"] == 1) and len(tr["value"]) == tr["count"]
    assert abs(sorted(tr["value"])[tr["count"] // 2] - b["value"]) <= 1e-9 * b["value"]
    # the copy bandwidth of the box, measured in this run
    assert 5500.0 < b["copy_peak_GBps_measured"] < 8000.0 and r["copy_peak_GBps_measured"] == b["copy_peak_GBps_measured"]
    assert abs(r["frac_of_measured_copy_peak"] - r["achieved"] / b["copy_peak_GBps_measured"]) < 1e-12
    assert r["copy_peak"]["copied_correctly"] is True
    modes = b["backward_modes"]
    assert modes["engine_one_launch_per_step"]["patches_per_s"] >= modes["engine_with_fill_and_scale_launches"]["patches_per_s"]
    if "secondary" in b:
        for k in ("K1_render_fwd", "K2_render_bwd", "K4_mix_materials", "K1_render_inputs_noise_clamp"):
            v = b["secondary"][k]
            assert v["frac_of_hbm_peak"] >= 0.70 and abs(v["frac_of_measured_copy_peak"] - v["algorithmic_GBps"] / b["copy_peak_GBps_measured"]) < 1e-9
        assert b["secondary"]["render_inputs_B8_views1"]["kernel_launches_per_call"] == 1.0
        assert b["secondary"]["LocalRenderer_render_host_tensor"]["renders_per_s"] > 2000.0
    c = b["cpu_baseline"]
    if c is not None:
        assert c["kind"] == "port" and c["unit"] == "patches/s" and c["value"] > 0 and 1 <= c["cores"] <= c["cpus"]
        assert isinstance(c["cpu_model"], str) and c["cpu_model"] and "patches" in c["sample"]
        assert all(isinstance(v, float) and v > 0 for v in c["probe_patches_per_s_by_threads"].values())
        assert isinstance(c["largest_probe_patches_per_s"], float)
        assert isinstance(c.get("all_cores_patches_per_s", 0.0), float)          # numeric or absent, never a sentence
        assert b["value"] / c["value"] > 1e3
        assert sum(b["wall_s"].values()) < 40.0                                  # the driver form is short


def test_bench_source_emits_the_contract_keys():
    with open(os.path.join(ROOT, "bench.py")) as f:
        src = f.read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "workload", "roofline", "bound", "achieved", "peak", "frac", "traffic",
                "cpu_baseline", "cores", "kind", "sample", "per_gpu_value", "value_through_autograd_engine", "valu_issue_frac",
                "all_cores_patches_per_s", "shader_cycles_per_launch", "valu_issue_frac"):
        assert '"%s"' % key in src, key


def test_short_form_reports_the_median_region():
    """bench.py's short forms time >= 9 regions and report the median one"""
    import bench
    assert bench.median_region_index([3.0]) == 0
    assert bench.median_region_index([5.0, 1.0, 3.0]) == 2
    t = [0.80, 0.76, 0.79, 0.77, 0.95, 0.78, 0.76, 0.81, 0.77]            # one outlier region
    assert t[bench.median_region_index(t)] == 0.78
    assert bench.median_region_index([1.0, 2.0, 3.0, 4.0]) == 2             # even count: the slower middle one
    with open(os.path.join(ROOT, "bench.py")) as f:
        src = f.read()
    assert "n_regions = 1 if args.steps >= 256 else max(9, args.regions)" in src


def _made_up_measurements(**over):
    B, steps = 8, 20
    regions = [0.00076, 0.00074, 0.00075, 0.00080, 0.00074, 0.00075, 0.00076, 0.00075, 0.00074]
    import bench
    med = bench.median_region_index(regions)
    m = {"B": B, "H": 256, "S": 9, "world": 1, "n_batches": 6, "elapsed": regions[med], "job_elapsed": regions,
         "median_region": med, "n_regions": 9, "kernel_ms": sorted([0.0381, 0.0379, 0.0385, 0.0380] * 8), "kernel_ms_avg": 0.038125,
         "region_ms_per_launch": 0.0361, "main_ns": 0, "clock_ghz": 2.2, "clock_note": "made up",
         "cycle_leg_ms": 0.0390, "other_ms_per_step": 0.0310, "other_ms": [0.060, 0.061, 0.062], "other_steps": 80,
         "leg_steps": 600, "engine_ms_per_step": 0.0356, "engine_plain_ms_per_step": 0.0400,
         "copy": {"GBps": 6550.0, "ms_per_launch": 0.328}, "mean_loss": 0.5,
         "per_rank": {"elapsed_s": [regions[med]], "cpus": ["0-63"], "pci_crosscheck": ["match"]}, "ranks_seen": 1,
         "process_group": None, "host_path": "native C++ extension (csrc/host_ext.cpp)"}
    m.update(over)
    return m


def _args(*argv):
    import sys
    import bench
    saved, sys.argv = sys.argv, ["bench.py"] + list(argv)
    try:
        return bench.parse_args()
    finally:
        sys.argv = saved


def test_line_assembly_one_clock_and_every_mode():
    """bench.assemble_line on made-up measurements (no GPU): the contract's identities in the default mode, under
    --timed-only (no follow-up leg ran: their figures are None, nothing raises) and for a --streams 2 timed region.
    (Round 6: a name clash in this code cost a whole profile collection -- it had only ever run on a GPU box.)"""
    import bench
    alg = 144.0 * 256 * 256 * 8
    out, copy_gbps = bench.assemble_line(_args("--steps", "20", "--warmup", "5"), _made_up_measurements())
    json.loads(json.dumps(out))                                             # serialisable
    r = out["roofline"]
    assert out["steps"] == 20 and abs(out["value"] - 8 * 1e3 / out["ms_per_step"]) <= 1e-9 * out["value"]
    assert out["value"] == out["value_through_autograd_engine"] and out["config"]["backward"] == "autograd engine"
    assert abs(out["backward_modes"]["engine_with_fill_and_scale_launches"]["patches_per_s"] - 8 / 0.0400e-3) < 1e-6
    assert copy_gbps == 6550.0 and "value_leaf_shortcut" not in out
    # ONE clock: frac x peak x ms_per_step / bytes == 1, exactly the interval `value` is priced with
    assert abs(r["frac"] * 8e12 * out["ms_per_step"] * 1e-3 / alg - 1.0) < 1e-12 and r["time_per_step_ms"] == out["ms_per_step"]
    assert abs(r["frac_by_launch_events"] - alg / 0.0361e-3 / 8e12) < 1e-12 and r["time_per_launch_ms"] == 0.0361
    assert abs(r["frac_of_measured_copy_peak"] - r["achieved"] / 6550.0) < 1e-12 and r["copy_peak_GBps_measured"] == 6550.0
    assert abs(r["shader_cycles_per_launch"] - 0.0390e-3 * 2.2e9) < 1e-6
    assert out["timed_regions"]["count"] == 9 and out["timed_regions"]["value"][out["timed_regions"]["median_index"]] == out["value"]
    if r["traffic"] is not None:                                            # the replayed counters of THIS build of the library
        assert alg <= r["traffic"] <= 1.05 * alg and 0.3 < r["valu_issue_frac"] < 1.0
    # --timed-only: what the profiler passes run
    quiet = _made_up_measurements(clock_ghz=None, clock_note="not measured (--timed-only)", cycle_leg_ms=None,
                                  other_ms_per_step=None, other_ms=[], other_steps=0, leg_steps=0,
                                  engine_ms_per_step=None, engine_plain_ms_per_step=None, copy=None, kernel_ms=[],
                                  kernel_ms_avg=None)
    out, copy_gbps = bench.assemble_line(_args("--steps", "20", "--timed-only"), quiet)
    json.loads(json.dumps(out))
    assert copy_gbps is None and out["follow_up_legs"].startswith("skipped") and out["value_two_streams_overlapped"] is None
    assert out["backward_modes"]["engine_one_launch_per_step"]["patches_per_s"] is None
    assert out["roofline"]["shader_cycles_per_launch"] is None and out["roofline"]["kernel_ms_median"] is None
    assert abs(out["roofline"]["frac"] * 8e12 * out["ms_per_step"] * 1e-3 / alg - 1.0) < 1e-12
    # a two-stream timed region: no single launch stream, so no event pair; the one clock still holds
    two = _made_up_measurements(main_ns=2, region_ms_per_launch=None)
    out, _ = bench.assemble_line(_args("--steps", "20", "--streams", "2"), two)
    json.loads(json.dumps(out))
    assert out["roofline"]["time_per_launch_ms"] is None and out["config"]["streams_per_gpu"] == 2
    assert abs(out["roofline"]["frac"] * 8e12 * out["ms_per_step"] * 1e-3 / alg - 1.0) < 1e-12
