"""Compile-time regression guard of the fused loss kernel (CPU test: hipcc cross-compiles gfx950 without a GPU).

The last ~10 % of K3's speed comes from per-translation-unit LLVM scheduler options and from algebra that removed
transcendentals; nothing at run time would notice a toolchain bump or an innocent edit undoing either.  This test
compiles the two forward+adjoint translation units to assembly with the Makefile's own flags and checks the headline
kernel k_rendering_loss_inl<GRAD=1,L1=0,HEAD=0> (and, more loosely, the MixedLoss / head-fused variants):

  * target gfx950, 4 waves/SIMD (<= 128 VGPRs), no AGPRs, no scratch at all in the headline kernel
  * tied-roughness scene loop: VALU count within the budget, exactly 13 transcendentals, no scratch traffic,
    no IEEE-division expansion (v_div_*), no packed math (the build uses -fno-slp-vectorize on purpose)
  * three-lobe (untied) scene loop: VALU and transcendental count, no scratch traffic
  * numerics contract: the products of the exact dot products on the coords -> NH path are never contracted into
    FMAs: every dot3 must appear as 3 v_mul + 2 v_add; checked on the stand-alone `svbrdf_isa_probe_dot3` kernel
    (n.wo, n.wi and wo.h, which do not feed NH, are explicit FMAs since round 4).
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
# a toolchain test: without hipcc there is nothing to compile (the GPU parity tests, by contrast, FAIL without a GPU)
pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs /opt/rocm/bin/hipcc (cross-compiles gfx950)")
CSRC = os.path.join(ROOT, "svbrdf_estimation_amd", "csrc")
HEADLINE = "k_rendering_loss_inlILb1ELb0ELb0"

# budgets: measured values of the shipped build + a small margin (tools/isa_stats.py prints the current ones).
# The scene loops hold TWO renders per trip (geometry ping-pong), so the counts are per two renders.
RENDERS_PER_TRIP = 2
TIED_LOOP_VALU_MAX = 618             # headline 602 = 301 per render (round 1: 336; round 2: 320; round 3: 323; round 4: n.wo / n.wi / wo.h as FMAs and the per-pixel gradient constants after the loop, f as d/pi + F (GD - d/pi)); device-table MixedLoss 610
TIED_LOOP_TRANS = 26                 # 13 per render
UNTIED_LOOP_VALU_MAX = 842           # RenderingLoss kernel: 826 (three lobes, channel by channel; round 3: 870)
UNTIED_LOOP_TRANS = 54               # 27 per render
UNTIED_EXTRA_VALU_MAX = {"mixed": 865, "head": 688}     # MixedLoss 848; head-fused 672 (input tied by construction)
UNTIED_EXTRA_TRANS = {"mixed": 54, "head": 42}
UNTIED_EXTRA_SCRATCH_MAX = {"mixed": 16, "head": 0}       # MixedLoss three-lobe loop: 13 spill accesses per trip
TIED_EXTRA_SCRATCH_MAX = {"mixed": 6, "head": 0}        # device-table MixedLoss kernel: 5 spill accesses per trip


def _make_var(name):
    out = subprocess.check_output(["make", "-s", "-C", CSRC, "print-" + name], text=True)
    return out.strip().split()


def _compile_tu(tmp, tu, sched_var):
    out = str(tmp / ("tu%d.s" % tu))
    cmd = (_make_var("HIPCC") + _make_var("HIPFLAGS") + _make_var(sched_var) +
           ["-DSVBRDF_TU=%d" % tu, "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "svbrdf_kernels.hip")])
    cmd = [c.replace("../../include", os.path.join(ROOT, "include")) for c in cmd]
    subprocess.check_call(cmd, cwd=CSRC, stderr=subprocess.DEVNULL)
    with open(out) as f:
        return f.read()


@pytest.fixture(scope="module")
def adjoint_asm(tmp_path_factory):
    """assembly of the RenderingLoss forward+adjoint translation unit (SVBRDF_TU=1), the Makefile's own flags"""
    return _compile_tu(tmp_path_factory.mktemp("isa1"), 1, "SCHED_ADJOINT")


@pytest.fixture(scope="module")
def adjoint_extra_asm(tmp_path_factory):
    """assembly of the MixedLoss / head-fused forward+adjoint translation unit (SVBRDF_TU=3)"""
    return _compile_tu(tmp_path_factory.mktemp("isa3"), 3, "SCHED_ADJOINT_EXTRA")


def test_headline_kernel_resources(adjoint_asm):
    import isa_stats
    assert '.amdgcn_target "amdgcn-amd-amdhsa--gfx950"' in adjoint_asm
    name, meta, whole, loops, ins, rng = isa_stats.analyse(adjoint_asm, HEADLINE)
    assert int(meta["NumVgprs"]) <= 128 and int(meta["NumAgprs"]) == 0, meta
    assert int(meta["Occupancy"]) >= 4, meta
    assert int(meta["ScratchSize"]) == 0 and whole["scratch"] == 0, (meta, whole)      # no spill anywhere in the kernel
    assert whole["v_div"] == 0 and whole["v_pk"] == 0, whole
    print("headline kernel: %s" % {k: meta[k] for k in ("NumVgprs", "TotalNumSgprs", "ScratchSize", "Occupancy")})


def test_plane_loads_are_issued_back_to_back(adjoint_asm, adjoint_extra_asm):
    """The 24 plane loads of a pixel (18-21 with the head fused) must be in flight together: the iterative-minreg
    scheduler likes to sink a load to its first use, and a prologue in which every load is followed by
    `s_waitcnt vmcnt(0)` (seen when the coordinate loads were moved in front of them) serialises 24 memory latencies
    in front of every wave's scene loop.  Checked: between the first and the last plane load of each forward+adjoint
    kernel there is no vector-memory wait at all."""
    import isa_stats
    for text in (adjoint_asm, adjoint_extra_asm):
        for k in isa_stats.kernels(text):
            if "k_rendering_loss" not in k:
                continue
            _, _, _, _, ins, rng = isa_stats.analyse(text, k)
            first_loop = min(a for a, _ in rng)
            loads = [i for i, (_, _, mn, ops) in enumerate(ins[:first_loop])
                     if mn and mn.startswith("buffer_load_dword")]
            assert len(loads) >= 18, (k, len(loads))
            # head-fused kernels decode the 9 encoded planes before they load the target: two groups there
            groups, start = [], loads[0]
            for a, b in zip(loads, loads[1:]):
                if any(m == "s_waitcnt" and "vmcnt" in o for _, _, m, o in ins[a:b] if m):
                    groups.append((start, a))
                    start = b
            groups.append((start, loads[-1]))
            assert len(groups) <= 2, "%s: plane loads split into %d groups by s_waitcnt vmcnt" % (k, len(groups))
            # The two coordinate loads must be issued IN FRONT of the plane loads (rendering_loss_body: one memory round trip
            # for everything instead of three in series).  Since round 5 they are ordinary loads pinned by a scheduling
            # barrier -- the compiler's own s_waitcnt bookkeeping covers them -- so only their position is checked here
            # (rounds 3-4 used inline asm the bookkeeping did not see, and this test also had to police register use).
            # (the by-value-table kernels, "_inl"; the device-table kernels load their coordinates the ordinary way)
            coords = [i for i, (_, _, mn, ops) in enumerate(ins[:first_loop]) if mn == "global_load_dword"]
            early = [i for i in coords if i < loads[0]]
            assert len(early) == (2 if "_inl" in k else 0), "%s: %d coordinate loads in front of the plane loads" % (k, len(early))


def test_no_inline_asm_memory_instructions_in_the_kernel_source():
    """VERDICT round 4 (fragility i): no load or store may be issued by inline asm outside the compiler's s_waitcnt
    bookkeeping.  The source may use asm only as an optimisation barrier (empty template)."""
    with open(os.path.join(CSRC, "svbrdf_kernels.hip")) as f:
        src = f.read()
    for m in re.finditer(r'asm\s+volatile\s*\(\s*"([^"]*)"', src):
        assert m.group(1) == "", "inline asm with instructions: %r" % m.group(1)


def test_scene_loops_instruction_budget(adjoint_asm):
    import isa_stats
    names = [k for k in isa_stats.kernels(adjoint_asm) if "k_rendering_loss" in k]
    assert len(names) == 2, names            # by-value scene table and device scene table
    for k in names:
        name, meta, whole, loops, ins, rng = isa_stats.analyse(adjoint_asm, k)
        assert len(loops) == 2, "expected the three-lobe and the tied scene loop, found %d loops" % len(loops)
        untied, tied = sorted(loops, key=lambda c: -c["valu"])
        print("tied loop: %s" % tied)
        print("untied loop: %s" % untied)
        assert tied["trans"] == TIED_LOOP_TRANS, tied
        assert tied["valu"] <= TIED_LOOP_VALU_MAX, tied
        # (the device-table kernel fetches the next renders' scene rows inside the loop; the by-value kernel reads them
        # with scalar loads from the kernel-argument segment)
        assert tied["scratch"] == 0 and tied["vmem"] <= (0 if "_inl" in k else 8) and tied["lds"] == 0, tied
        assert tied["v_div"] == 0 and tied["v_pk"] == 0, tied
        assert untied["trans"] == UNTIED_LOOP_TRANS, untied
        assert untied["valu"] <= UNTIED_LOOP_VALU_MAX, untied
        assert untied["scratch"] == 0 and untied["v_div"] == 0, untied


def test_mixed_and_head_fused_variants_budget(adjoint_extra_asm):
    import isa_stats
    names = [k for k in isa_stats.kernels(adjoint_extra_asm) if "k_rendering_loss" in k]
    assert len(names) == 6, names            # {by-value, device table} x {L1 only, HEAD only, HEAD + L1}
    for k in names:
        _, meta, whole, loops, _, _ = isa_stats.analyse(adjoint_extra_asm, k)
        assert int(meta["NumVgprs"]) <= 128 and int(meta["Occupancy"]) >= 4, (k, meta)
        assert whole["v_div"] == 0 and whole["v_pk"] == 0, k
        kind = "head" if k.endswith("Lb1EEEvNS_10SceneBlockEPKfS3_S3_ffdfNS_8L1ParamsEPfPyS5_iii") or "ELb1EEEvPKf" in k else "mixed"
        tied = min((c for c in loops if c["trans"] == TIED_LOOP_TRANS), key=lambda c: c["valu"])
        untied = max(loops, key=lambda c: c["valu"])
        assert tied["scratch"] <= TIED_EXTRA_SCRATCH_MAX[kind] and tied["valu"] <= TIED_LOOP_VALU_MAX, (k, tied)
        assert untied["trans"] == UNTIED_EXTRA_TRANS[kind] and untied["valu"] <= UNTIED_EXTRA_VALU_MAX[kind], (k, kind, untied)
        assert untied["scratch"] <= UNTIED_EXTRA_SCRATCH_MAX[kind], (k, kind, untied)


def test_dot_products_are_not_contracted(tmp_path):
    """numerics contract (svbrdf_kernels.hip header): dot3 = three separately rounded products summed (p0+p1)+p2.
    The probe kernel is dot3 and nothing else; with the product build flags it must compile to 3 v_mul + 2 v_add
    and no FMA (an -ffp-contract=fast toolchain default would fuse two of the products)."""
    import isa_stats
    src = tmp_path / "probe.hip"
    src.write_text('#define SVBRDF_ISA_PROBE 1\n#include "%s"\n' % os.path.join(CSRC, "svbrdf_kernels.hip"))
    out = str(tmp_path / "probe.s")
    cmd = (_make_var("HIPCC") + _make_var("HIPFLAGS") + _make_var("SCHED_MAIN") +
           ["-DSVBRDF_TU=0", "-S", "--cuda-device-only", "-o", out, str(src)])
    cmd = [c.replace("../../include", os.path.join(ROOT, "include")) for c in cmd]
    subprocess.check_call(cmd, cwd=CSRC, stderr=subprocess.DEVNULL)
    _, _, whole, _, ins, _ = isa_stats.analyse(open(out).read(), "svbrdf_isa_probe_dot3")
    mns = [mn for _, _, mn, _ in ins if mn and mn.startswith("v_") and not mn.startswith("v_mov")]
    muls = [m for m in mns if m.startswith("v_mul_f32")]
    adds = [m for m in mns if m.startswith("v_add_f32")]
    assert len(muls) == 3 and len(adds) == 2 and whole["fma"] == 0, mns
