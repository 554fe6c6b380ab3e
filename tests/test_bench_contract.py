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
