"""ctypes binding of oracle/svbrdf_oracle.c (TEST INFRASTRUCTURE, not product).

Every function takes/returns C-contiguous numpy arrays.  The restated reference
lines are cited in svbrdf_core.inc.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVBRDF_ORACLE_SO: another build of the same source (the -fsanitize=address,undefined build of `make asan`, loaded by
# tests/test_sanitizers.py into a child interpreter with the ASan runtime preloaded)
_SO = os.environ.get("SVBRDF_ORACLE_SO") or os.path.join(_HERE, "_build", "libsvbrdf_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, n) for n in ("svbrdf_oracle.c", "svbrdf_core.inc", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.svbrdf_oracle_max_threads.restype = ctypes.c_int
    return _lib


def _p(a, ty=_f32p):
    return a.ctypes.data_as(ty)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _dims(maps, scenes):
    maps = _f32(maps)
    scenes = _f32(scenes)
    assert maps.ndim == 4 and maps.shape[1] == 12, maps.shape
    B, _, H, W = maps.shape
    assert scenes.ndim == 3 and scenes.shape[0] == B and scenes.shape[2] == 9, scenes.shape
    return maps, scenes, B, scenes.shape[1], H, W


def set_threads(n):
    lib().svbrdf_oracle_set_threads(int(n))


def max_threads():
    return int(lib().svbrdf_oracle_max_threads())


def make_xrow(W):
    x = np.empty(W, dtype=np.float32)
    lib().svbrdf_oracle_make_xrow(_p(x), ctypes.c_int(W))
    return x


def render_fwd(maps, scenes, xrow=None, f64=False):
    """maps [B,12,H,W], scenes [B,S,9] (cam xyz | light xyz | light rgb) -> [B,S,3,H,W]."""
    maps, scenes, B, S, H, W = _dims(maps, scenes)
    xrow = make_xrow(W) if xrow is None else _f32(xrow)
    out = np.empty((B, S, 3, H, W), dtype=np.float64 if f64 else np.float32)
    fn = lib().svbrdf_oracle_render_fwd_f64 if f64 else lib().svbrdf_oracle_render_fwd
    rc = fn(_p(maps), _p(scenes), _p(xrow), _p(out, _f64p if f64 else _f32p), B, S, H, W)
    if rc:
        raise RuntimeError("oracle render_fwd rc=%d" % rc)
    return out


def render_bwd(maps, scenes, grad_out, xrow=None, f64=False):
    """adjoint of render_fwd: grad_out [B,S,3,H,W] -> grad_maps [B,12,H,W]."""
    maps, scenes, B, S, H, W = _dims(maps, scenes)
    xrow = make_xrow(W) if xrow is None else _f32(xrow)
    grad_out = _f32(grad_out)
    assert grad_out.shape == (B, S, 3, H, W)
    gm = np.empty((B, 12, H, W), dtype=np.float64 if f64 else np.float32)
    fn = lib().svbrdf_oracle_render_bwd_f64 if f64 else lib().svbrdf_oracle_render_bwd
    rc = fn(_p(maps), _p(scenes), _p(xrow), _p(grad_out), _p(gm, _f64p if f64 else _f32p), B, S, H, W)
    if rc:
        raise RuntimeError("oracle render_bwd rc=%d" % rc)
    return gm


def rendering_loss(input, target, scenes, eps=0.1, xrow=None, want_grad=True, f64=False):
    """losses.py:29-52 with explicit scenes.  Returns (loss: float, grad_input or None)."""
    input, scenes, B, S, H, W = _dims(input, scenes)
    target = _f32(target)
    assert target.shape == input.shape
    xrow = make_xrow(W) if xrow is None else _f32(xrow)
    loss = ctypes.c_double(0.0)
    gty = np.float64 if f64 else np.float32
    grad = np.empty((B, 12, H, W), dtype=gty) if want_grad else None
    fn = lib().svbrdf_oracle_rendering_loss_f64 if f64 else lib().svbrdf_oracle_rendering_loss
    gp = _p(grad, _f64p if f64 else _f32p) if want_grad else None
    rc = fn(_p(input), _p(target), _p(scenes), _p(xrow), ctypes.c_float(eps),
            ctypes.byref(loss), gp, B, S, H, W)
    if rc:
        raise RuntimeError("oracle rendering_loss rc=%d" % rc)
    return loss.value, grad


def mixed_loss(input, target, scenes, l1_weight=0.1, eps=0.1, eps_l1=0.01, xrow=None, want_grad=True, f64=False):
    """losses.py:54-63: l1_weight * SVBRDFL1Loss (losses.py:7-19) + RenderingLoss, explicit scenes."""
    input, scenes, B, S, H, W = _dims(input, scenes)
    target = _f32(target)
    assert target.shape == input.shape
    xrow = make_xrow(W) if xrow is None else _f32(xrow)
    loss = ctypes.c_double(0.0)
    grad = np.empty((B, 12, H, W), dtype=np.float64 if f64 else np.float32) if want_grad else None
    fn = lib().svbrdf_oracle_mixed_loss_f64 if f64 else lib().svbrdf_oracle_mixed_loss
    gp = _p(grad, _f64p if f64 else _f32p) if want_grad else None
    rc = fn(_p(input), _p(target), _p(scenes), _p(xrow), ctypes.c_float(eps), ctypes.c_float(l1_weight),
            ctypes.c_float(eps_l1), ctypes.byref(loss), gp, B, S, H, W)
    if rc:
        raise RuntimeError("oracle mixed_loss rc=%d" % rc)
    return loss.value, grad


def head_decode(encoded9):
    """models.py:338-346: [B,9,H,W] post-tanh generator output -> [B,12,H,W] maps."""
    enc = _f32(encoded9)
    assert enc.ndim == 4 and enc.shape[1] == 9
    B, _, H, W = enc.shape
    maps = np.empty((B, 12, H, W), dtype=np.float32)
    rc = lib().svbrdf_oracle_head_decode(_p(enc), _p(maps), B, H, W)
    if rc:
        raise RuntimeError("oracle head_decode rc=%d" % rc)
    return maps


def head_loss(encoded9, target, scenes, l1_weight=0.1, eps=0.1, eps_l1=0.01, xrow=None, want_grad=True, f64=False):
    """mixed loss of the decoded head output; gradient w.r.t. the 9 encoded channels (float64 array)."""
    enc = _f32(encoded9)
    target, scenes, B, S, H, W = _dims(target, scenes)
    assert enc.shape == (B, 9, H, W)
    xrow = make_xrow(W) if xrow is None else _f32(xrow)
    loss = ctypes.c_double(0.0)
    grad = np.empty((B, 9, H, W), dtype=np.float64) if want_grad else None
    rc = lib().svbrdf_oracle_head_loss(_p(enc), _p(target), _p(scenes), _p(xrow), ctypes.c_float(eps),
                                       ctypes.c_float(l1_weight), ctypes.c_float(eps_l1), ctypes.byref(loss),
                                       _p(grad, _f64p) if want_grad else None, int(bool(f64)), B, S, H, W)
    if rc:
        raise RuntimeError("oracle head_loss rc=%d" % rc)
    return loss.value, grad


def loss_tie_map(input, target, scenes, eps=0.1, xrow=None):
    """[B,H,W] float64: smallest |log difference| over scenes/channels per pixel (see svbrdf_oracle.c)."""
    input, scenes, B, S, H, W = _dims(input, scenes)
    target = _f32(target)
    xrow = make_xrow(W) if xrow is None else _f32(xrow)
    out = np.empty((B, H, W), dtype=np.float64)
    rc = lib().svbrdf_oracle_loss_tie_map(_p(input), _p(target), _p(scenes), _p(xrow), ctypes.c_float(eps),
                                          _p(out, _f64p), B, S, H, W)
    if rc:
        raise RuntimeError("oracle loss_tie_map rc=%d" % rc)
    return out
