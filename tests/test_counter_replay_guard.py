"""bench.py's roofline.traffic and VALU instruction counts are REPLAYED from
profiles/k3_hbm_traffic.json (rocprofv3 --pmc passes of an earlier run), so the record must be keyed to the kernel it was
taken from -- the sha256 of the kernel's instruction bytes in the shipped library -- and refused for any other code.
CPU only: hipcc cross-compiles, the hash is read out of the .so's offload bundle (svbrdf_estimation_amd/_codehash.py)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svbrdf_estimation_amd", "csrc")
LIB = os.path.join(ROOT, "svbrdf_estimation_amd", "lib", "libsvbrdf_hip.so")
RECORD = os.path.join(ROOT, "profiles", "k3_hbm_traffic.json")


def _make_var(name):
    return subprocess.check_output(["make", "-s", "-C", CSRC, "print-" + name], text=True).strip().split()


def _build_adjoint_unit(tmp, tag, extra):
    """the RenderingLoss forward+adjoint translation unit alone, Makefile flags + `extra`, linked into a .so of its own (the
    hasher only needs the unit's offload bundle)"""
    obj, so = str(tmp / (tag + ".o")), str(tmp / ("lib" + tag + ".so"))
    cmd = (_make_var("HIPCC") + _make_var("HIPFLAGS") + _make_var("SCHED_ADJOINT") + ["-DSVBRDF_TU=1"] + extra +
           ["-c", "-o", obj, os.path.join(CSRC, "svbrdf_kernels.hip")])
    cmd = [c.replace("../../include", os.path.join(ROOT, "include")) for c in cmd]
    subprocess.check_call(cmd, cwd=CSRC, stderr=subprocess.DEVNULL)
    subprocess.check_call(_make_var("HIPCC") + ["--offload-arch=gfx950", "-fPIC", "-shared", "-o", so, obj], stderr=subprocess.DEVNULL)
    return so


def test_record_is_keyed_to_the_shipped_kernel_and_replayed_for_it():
    import bench
    from svbrdf_estimation_amd import _codehash
    with open(RECORD) as f:
        rec = json.load(f)
    h = _codehash.k3_headline_hash(LIB)
    assert "k_rendering_loss_inl" in h["symbol"] and "ILb1ELb0ELb0EE" in h["symbol"] and h["bytes"] > 4000
    # the shipped kernel IS the code the counters were recorded on; if this fails the kernel changed: re-record the counters
    # (tools/collect_profiles.sh on the GPU box, tools/summarize_profiles.py here) -- bench.py reports traffic: null until then
    assert rec["kernel_code_sha256"] == h["sha256"], "K3 changed since profiles/k3_hbm_traffic.json was recorded"
    traffic, source, used = bench.replayed_counters(LIB, rec["B"], rec["H"], rec["S"])
    assert traffic == rec["hbm_bytes_per_launch"] and used["valu_wave_instr_per_launch"] == rec["valu_wave_instr_per_launch"]
    assert "NOT measured in this run" in source and h["sha256"][:16] in source
    # another shape is not this record's either
    assert bench.replayed_counters(LIB, rec["B"], 512, rec["S"])[0] is None


def test_replay_is_refused_for_any_other_build_of_the_kernel(tmp_path):
    """flip ONE constant of K3 (the first-round load stagger, 64 -> 48 sleep units) in a scratch build: different machine
    code, so the recorded counters are not its counters -- no traffic, no VALU counts, and the line says why.  An unchanged
    rebuild hashes like the shipped library (the hash covers instruction bytes only: no build ids, no paths)."""
    import bench
    from svbrdf_estimation_amd import _codehash
    with open(RECORD) as f:
        rec = json.load(f)
    same = _build_adjoint_unit(tmp_path, "same", [])
    assert _codehash.k3_headline_hash(same)["sha256"] == _codehash.k3_headline_hash(LIB)["sha256"]
    other = _build_adjoint_unit(tmp_path, "flipped", ["-DSVBRDF_K3_STAGGER=48"])
    assert _codehash.k3_headline_hash(other)["sha256"] != rec["kernel_code_sha256"]
    traffic, source, used = bench.replayed_counters(other, rec["B"], rec["H"], rec["S"])
    assert traffic is None and used is None
    assert "NOT replayed" in source and "re-record" in source and rec["kernel_code_sha256"][:16] in source
    # a record without a hash (the pre-round-5 file) is never replayed
    old = dict(rec)
    del old["kernel_code_sha256"]
    p = tmp_path / "old.json"
    p.write_text(json.dumps(old))
    assert bench.replayed_counters(LIB, rec["B"], rec["H"], rec["S"], str(p))[0] is None
    with pytest.raises(LookupError):
        _codehash.kernel_code_sha256(LIB, ("k_no_such_kernel",))
