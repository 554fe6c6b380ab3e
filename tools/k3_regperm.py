"""Research tool (register assignment): rewrite the VGPR numbers of ONE kernel in the compiler's
assembly output by a permutation of aligned register PAIRS (64-bit tuples stay valid), assemble and link a code object.
   python tools/k3_regperm.py base.s out_dir N [seed]        -> out_dir/p<i>.hsaco + p<i>.perm for N random permutations
   python tools/k3_regperm.py base.s out_dir --perm file     -> one code object from a stored permutation
Pairs 0..2 (v0-v5: work-item id at entry, the one 4-tuple) are never moved."""
import os, random, re, subprocess, sys

KERNEL = "_ZN12_GLOBAL__N_116k_rendering_lossILb1ELb0ELb0EEEvPKfS2_S2_S2_ffdfNS_8L1ParamsEPfPyS4_iii"
CLANG, LLD = "/opt/rocm/lib/llvm/bin/clang", "/opt/rocm/lib/llvm/bin/ld.lld"
FIXED = 3


def split(text):
    lines = text.split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    return lines[:a], lines[a:b + 1], lines[b + 1:]


def rename(body, pair_perm):
    def reg(r):
        return 2 * pair_perm[r // 2] + (r & 1)

    def one(m):
        return "v%d" % reg(int(m.group(1)))

    def rng(m):
        lo, hi = int(m.group(1)), int(m.group(2))
        nlo = reg(lo)
        assert [reg(r) for r in range(lo, hi + 1)] == list(range(nlo, nlo + hi - lo + 1)), (lo, hi)
        return "v[%d:%d]" % (nlo, nlo + hi - lo)

    out = []
    for l in body:
        code, sep, comment = l.partition(";")
        code = re.sub(r"\bv\[(\d+):(\d+)\]", rng, code)
        code = re.sub(r"\bv(\d+)\b", one, code)
        out.append(code + sep + comment)
    return out


def build(head, body, tail, pair_perm, path):
    s = path + ".s"
    open(s, "w").write("\n".join(head + rename(body, pair_perm) + tail))
    subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", path + ".o"])
    subprocess.check_call([LLD, "-shared", path + ".o", "-o", path + ".hsaco"])
    os.remove(s); os.remove(path + ".o")
    open(path + ".perm", "w").write(" ".join(map(str, pair_perm)))


def neighbours(base, n, rnd, swaps):
    for _ in range(n):
        p = list(base)
        for _ in range(swaps):
            i, j = rnd.sample(range(FIXED, 64), 2)
            p[i], p[j] = p[j], p[i]
        yield p


if __name__ == "__main__":
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    head, body, tail = split(open(src).read())
    ident = list(range(64))
    if sys.argv[3] == "--perm":
        build(head, body, tail, list(map(int, open(sys.argv[4]).read().split())), os.path.join(out, "tuned"))
    else:
        n, seed = int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0
        base = list(map(int, open(sys.argv[5]).read().split())) if len(sys.argv) > 5 else ident
        swaps = int(sys.argv[6]) if len(sys.argv) > 6 else 61
        rnd = random.Random(seed)
        build(head, body, tail, base, os.path.join(out, "p000"))
        for i, p in enumerate(neighbours(base, n, rnd, swaps), 1):
            build(head, body, tail, p, os.path.join(out, "p%03d" % i))
