"""Instruction statistics of a kernel in hipcc's assembly output (`hipcc -S --cuda-device-only`).

Used by tests/test_isa_guard.py (compile-time regression guard of the fused loss kernel: the last ~10 % of its
speed rests on scheduler options a toolchain bump could silently undo) and by hand:

    python tools/isa_stats.py file.s [kernel-name-substring]

A "loop" is the instruction range between a label and the last backward branch to it; nested ranges are
reported separately (the scene loops of K3 contain no inner loops).
"""
import re
import sys

TRANSCENDENTAL = ("v_rcp_", "v_rsq_", "v_log_", "v_exp_", "v_sqrt_", "v_sin_", "v_cos_")


def kernels(text):
    """{symbol: (list of lines of the body, metadata dict)} for every function in the file"""
    out = {}
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", lines[i])
        if m and not lines[i].startswith(".L"):
            name = m.group(1)
            j = i + 1
            body = []
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                body.append(lines[j])
                j += 1
            meta = {}
            k = j
            while k < len(lines) and k < j + 120:
                mm = re.match(r"^\s*\.amdhsa_(\w+)\s+(\S+)", lines[k])
                if mm:
                    meta[mm.group(1)] = mm.group(2)
                mm = re.match(r"^;\s*(\w[\w ]*\w):\s*(\S+)", lines[k])
                if mm:
                    meta[mm.group(1)] = mm.group(2)
                if lines[k].startswith("\t.end_amdhsa_kernel") or (k > j + 5 and re.match(r"^(_Z\w+):", lines[k])):
                    pass
                k += 1
            out[name] = (body, meta)
            i = j
        else:
            i += 1
    return out


def instructions(body):
    """[(index_in_body, label_or_None, mnemonic, operands)]"""
    res = []
    for idx, line in enumerate(body):
        s = line.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            res.append((idx, m.group(1), None, None))
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        res.append((idx, None, parts[0], parts[1] if len(parts) > 1 else ""))
    return res


def loops(ins):
    """[(start, end)] instruction-list index ranges [label position, backward branch position]"""
    pos = {lab: i for i, (_, lab, mn, _) in enumerate(ins) if lab}
    found = {}
    for i, (_, lab, mn, ops) in enumerate(ins):
        if mn and mn.startswith(("s_cbranch", "s_branch")):
            tgt = ops.strip()
            if tgt in pos and pos[tgt] < i:
                found[pos[tgt]] = max(found.get(pos[tgt], 0), i)
    return sorted(found.items())


def classify(ins_slice):
    c = {"total": 0, "valu": 0, "trans": 0, "fma": 0, "salu": 0, "smem": 0, "vmem": 0, "scratch": 0, "lds": 0,
         "v_div": 0, "v_pk": 0, "cndmask": 0, "cmp": 0, "waitcnt": 0, "mov": 0, "sgpr_src_valu": 0}
    for _, lab, mn, ops in ins_slice:
        if not mn:
            continue
        c["total"] += 1
        if mn.startswith("v_"):
            c["valu"] += 1
            if mn.startswith(TRANSCENDENTAL):
                c["trans"] += 1
            if mn.startswith(("v_fma_", "v_fmac_", "v_mad_", "v_mac_")) or mn.startswith("v_pk_fma"):
                c["fma"] += 1
            if mn.startswith("v_div_"):
                c["v_div"] += 1
            if mn.startswith("v_pk_"):
                c["v_pk"] += 1
            if mn.startswith("v_cndmask"):
                c["cndmask"] += 1
            if mn.startswith("v_cmp"):
                c["cmp"] += 1
            if mn.startswith("v_mov_") or mn.startswith("v_accvgpr"):
                c["mov"] += 1
            srcs = ops.split(",")[1:] if not mn.startswith("v_cmp") else ops.split(",")
            if any(re.match(r"^\s*-?\|?s\d+|^\s*-?\|?s\[", x) for x in srcs):
                c["sgpr_src_valu"] += 1
        elif mn.startswith("scratch_"):
            c["scratch"] += 1
        elif mn.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif mn.startswith("ds_"):
            c["lds"] += 1
        elif mn.startswith("s_load") or mn.startswith("s_buffer_load"):
            c["smem"] += 1
        elif mn.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif mn.startswith("s_"):
            c["salu"] += 1
    return c


def analyse(text, name_substring):
    """(symbol, meta, whole-kernel counts, [(loop counts)] sorted by position) of the single kernel matching"""
    ks = {k: v for k, v in kernels(text).items() if name_substring in k}
    if len(ks) != 1:
        raise KeyError("%d kernels match %r: %s" % (len(ks), name_substring, sorted(ks)))
    (name, (body, meta)), = ks.items()
    ins = instructions(body)
    return name, meta, classify(ins), [classify(ins[a:b + 1]) for a, b in loops(ins)], ins, loops(ins)


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in sorted(kernels(text)):
        if sub not in k:
            continue
        name, meta, whole, ls, ins, rng = analyse(text, k)
        print(name)
        print("   meta:", {x: meta[x] for x in ("next_free_vgpr", "next_free_sgpr", "ScratchSize", "Occupancy") if x in meta})
        print("   whole:", whole)
        for (a, b), c in zip(rng, ls):
            print("   loop @%d..%d:" % (a, b), c)
