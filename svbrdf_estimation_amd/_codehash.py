"""Identity of a kernel's MACHINE CODE inside libsvbrdf_hip.so (no GPU, no external tools).

bench.py replays hardware-counter figures (HBM bytes, VALU instruction counts per launch) that were recorded with
``rocprofv3 --pmc`` in a separate profiling run (profiles/k3_hbm_traffic.json).  Such a replay is only honest for the very
code the counters were taken from, so the record carries the sha256 of the fused-loss kernel's instruction bytes and
bench.py compares it with the library it is about to run: any change to the kernel -- source, flags, scheduler options,
compiler -- changes the hash, and the replay is refused (``traffic: null`` with a note) until the counters are re-recorded.

What is hashed: the bytes of the kernel's function symbol in the gfx950 code object -- the ``.text`` range
[st_value, st_value + st_size) of the ELF that sits in the library's ``.hip_fatbin`` clang offload bundle.  Instruction
bytes only: no kernel descriptor, no metadata, no build ids, so rebuilding unchanged source gives the same hash.
"""
import hashlib
import struct

_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_objects(so_bytes, arch="gfx950"):
    """the device ELFs for `arch` in every (uncompressed) clang offload bundle of a host shared library"""
    out, pos = [], 0
    while True:
        off = so_bytes.find(_BUNDLE_MAGIC, pos)
        if off < 0:
            return out
        pos = off + len(_BUNDLE_MAGIC)
        n, = struct.unpack_from("<Q", so_bytes, off + 24)
        if n > 64:          # the magic string inside some other data
            continue
        q = off + 32
        for _ in range(n):
            eo, es, ts = struct.unpack_from("<QQQ", so_bytes, q)
            q += 24
            triple = so_bytes[q:q + ts].decode("ascii", "replace")
            q += ts
            if triple.startswith("hip") and triple.rstrip("-").endswith(arch) and es > 0:
                elf = so_bytes[off + eo:off + eo + es]
                if elf[:4] == b"\x7fELF":
                    out.append(elf)


def function_symbols(elf):
    """{name: instruction bytes} for the FUNC symbols of a 64-bit little-endian ELF"""
    if elf[:6] != b"\x7fELF\x02\x01":
        raise ValueError("not a 64-bit little-endian ELF")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        (_name, typ, _flags, addr, off, size, link, _info, _align, entsize) = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append((typ, addr, off, size, link, entsize))
    out = {}
    for typ, _addr, off, size, link, entsize in secs:
        if typ != 2:            # SHT_SYMTAB
            continue
        str_off = secs[link][2]
        for k in range(size // (entsize or 24)):
            name, info, _other, shndx, value, sz = struct.unpack_from("<IBBHQQ", elf, off + k * 24)
            if (info & 0xF) != 2 or sz == 0 or shndx == 0 or shndx >= len(secs):       # STT_FUNC, defined
                continue
            end = elf.index(b"\0", str_off + name)
            sec_typ, sec_addr, sec_off, _s, _l, _e = secs[shndx]
            if sec_typ == 8:    # SHT_NOBITS
                continue
            start = sec_off + (value - sec_addr)
            out[elf[str_off + name:end].decode("ascii", "replace")] = elf[start:start + sz]
    return out


def kernel_code_sha256(so_path, name_substrings):
    """sha256 (hex) of the instruction bytes of the ONE kernel whose mangled name contains every string in
    `name_substrings`, with its mangled name and size: {"sha256", "symbol", "bytes"}.  Raises LookupError when no such
    kernel exists in the library or when the strings do not single one out."""
    with open(so_path, "rb") as f:
        data = f.read()
    hits = {}
    for elf in device_code_objects(data):
        for name, code in function_symbols(elf).items():
            if all(s in name for s in name_substrings):
                hits[name] = code
    if len(hits) != 1:
        raise LookupError("%d kernels match %r in %s: %s" % (len(hits), name_substrings, so_path, sorted(hits)[:4]))
    (name, code), = hits.items()
    return {"sha256": hashlib.sha256(code).hexdigest(), "symbol": name, "bytes": len(code)}


# the headline kernel of bench.py: k_rendering_loss_inl<GRAD=true, L1=false, HEAD=false> (scene table by value)
K3_HEADLINE = ("k_rendering_loss_inl", "ILb1ELb0ELb0EE")


def k3_headline_hash(so_path=None):
    if so_path is None:
        from . import _native
        so_path = _native.library_path()
    return kernel_code_sha256(so_path, K3_HEADLINE)
