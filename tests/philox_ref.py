"""Test infrastructure: numpy restatement of the counter-based normal field of ``svbrdf_render_inputs`` (include/svbrdf_hip.h)
-- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the published algorithm,
pinned here by the known-answer vectors of the Random123 distribution, tests/test_host_logic.py) and the Box-Muller map the
header states.  Only tests import this; the product draws its noise in the HIP kernel."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """counter words (uint32 arrays or scalars, broadcast) and key -> four uint32 arrays"""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                       # 32 x 32 -> 64 bit products
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def normal_field(seed, offset, n_elements):
    """float64 standard normals of output elements 0 .. n_elements-1 for (seed, offset), as the header defines them"""
    groups = (n_elements + 3) // 4
    g = np.arange(groups, dtype=np.uint64)
    x = philox4x32_10(g & MASK, g >> np.uint64(32), offset & 0xFFFFFFFF, (offset >> 32) & 0xFFFFFFFF,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [((w >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0 for w in x]
    out = np.empty((groups, 4), dtype=np.float64)
    for h in range(2):
        r = np.sqrt(-2.0 * np.log(u[2 * h]))
        out[:, 2 * h] = r * np.cos(2.0 * np.pi * u[2 * h + 1])
        out[:, 2 * h + 1] = r * np.sin(2.0 * np.pi * u[2 * h + 1])
    return out.reshape(-1)[:n_elements]
