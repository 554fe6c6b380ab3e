"""Deterministic, platform-independent synthetic SVBRDF maps for tests and fixtures.

Only integer hashing, IEEE add/mul/div and correctly rounded sqrt are used, so the
fp32 arrays are bit-identical in the build container (where the golden fixtures are
generated from the reference) and on the GPU box (where they are regenerated and
fed to the HIP kernels).  Fixtures store a checksum of the regenerated inputs.
"""
import hashlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, shape):
    """float32 uniform in [0, 1) with 24 random mantissa bits (counter based)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
    bits = _splitmix64(_splitmix64(ctr)) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(shape)


def approx_normal(seed, shape):
    """Irwin-Hall(4) centred and scaled to unit variance: adds and one multiply only."""
    u = uniform01(seed, (4,) + tuple(shape))
    s = ((u[0] + u[1]) + (u[2] + u[3])) - np.float32(2.0)
    return s * np.float32(1.7320508)


def make_maps(seed, B, H, W=None, tilt=0.3, r_lo=0.0, r_hi=1.0, tiled_roughness=True, unit_normals=True):
    """[B,12,H,W] float32: normals | diffuse | roughness | specular (utils.py:36-58 order)."""
    W = H if W is None else W
    t = np.float32(tilt)
    nx = t * approx_normal(seed * 16 + 1, (B, 1, H, W))
    ny = t * approx_normal(seed * 16 + 2, (B, 1, H, W))
    nz = np.float32(1.0) + np.abs(t * approx_normal(seed * 16 + 3, (B, 1, H, W)))
    n = np.concatenate([nx, ny, nz], axis=1)
    if unit_normals:
        n = n / np.sqrt((n[:, 0:1] * n[:, 0:1] + n[:, 1:2] * n[:, 1:2]) + n[:, 2:3] * n[:, 2:3])
    d = uniform01(seed * 16 + 4, (B, 3, H, W))
    if tiled_roughness:
        r = np.repeat(uniform01(seed * 16 + 5, (B, 1, H, W)), 3, axis=1)
    else:
        r = uniform01(seed * 16 + 5, (B, 3, H, W))
    r = np.float32(r_lo) + np.float32(r_hi - r_lo) * r
    s = uniform01(seed * 16 + 6, (B, 3, H, W))
    return np.ascontiguousarray(np.concatenate([n, d, r, s], axis=1), dtype=np.float32)


def checksum(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
