"""ctypes binding of libsvbrdf_hip.so (C ABI: include/svbrdf_hip.h).

PyTorch is used here only for device memory, the current HIP stream and (elsewhere)
autograd glue.  No arithmetic of the hot path happens in Python.  If the shared
library is missing or fails to load, every entry point raises NativeLibraryError --
there is deliberately no fallback.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVBRDF_HIP_LIB: load another build of the same ABI (ablation/experiment builds of tools/); default in-tree
_SO = os.environ.get("SVBRDF_HIP_LIB") or os.path.join(_HERE, "lib", "libsvbrdf_hip.so")
ABI_VERSION = 7

_lock = threading.Lock()
_lib = None
_xrow_cache = {}
_workspace_cache = {}
_launch_hook = None


def set_launch_hook(fn):
    """Measurement aid: ``fn("begin")`` / ``fn("end")`` is called immediately before / after the
    kernel launches of the fused loss are enqueued (bench.py records HIP events there).
    None disables it.  No effect on results."""
    global _launch_hook
    _launch_hook = fn


class NativeLibraryError(RuntimeError):
    """libsvbrdf_hip.so is missing, stale or reported an error."""


def library_path():
    return _SO


_fp = ctypes.c_void_p


def _load():
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_SO):
            raise NativeLibraryError(
                "HIP extension not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C svbrdf_estimation_amd/csrc`. There is no CPU fallback." % _SO)
        try:
            # torch is imported first on purpose: its bundled libamdhip64.so.7 is then the
            # one HIP runtime of the process and our NEEDED entry resolves to it by SONAME,
            # so the stream handles torch hands us belong to the runtime that launches.
            lib = ctypes.CDLL(_SO)
        except OSError as e:  # pragma: no cover
            raise NativeLibraryError("cannot load %s: %s" % (_SO, e))
        lib.svbrdf_abi_version.restype = ctypes.c_int
        lib.svbrdf_last_error.restype = ctypes.c_char_p
        lib.svbrdf_make_xrow.argtypes = [_fp, ctypes.c_int]
        lib.svbrdf_render_fwd.argtypes = [_fp, _fp, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
        lib.svbrdf_render_bwd.argtypes = [_fp, _fp, _fp, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
        lib.svbrdf_rendering_loss_workspace_bytes.argtypes = [ctypes.c_int] * 4
        lib.svbrdf_rendering_loss_workspace_bytes.restype = ctypes.c_size_t
        lib.svbrdf_rendering_loss_fwd_bwd.argtypes = (
            [_fp, _fp, _fp, _fp, ctypes.c_float, _fp, _fp, _fp, ctypes.c_size_t] + [ctypes.c_int] * 4 + [_fp])
        lib.svbrdf_mixed_loss_fwd_bwd.argtypes = (
            [_fp, _fp, _fp, _fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp, ctypes.c_size_t]
            + [ctypes.c_int] * 4 + [_fp])
        lib.svbrdf_mixed_loss_fwd_bwd.restype = ctypes.c_int
        lib.svbrdf_head_loss_fwd_bwd.argtypes = lib.svbrdf_mixed_loss_fwd_bwd.argtypes
        lib.svbrdf_head_loss_fwd_bwd.restype = ctypes.c_int
        for name in ("svbrdf_mixed_loss_fwd_bwd_host_scenes", "svbrdf_head_loss_fwd_bwd_host_scenes"):
            getattr(lib, name).argtypes = lib.svbrdf_mixed_loss_fwd_bwd.argtypes
            getattr(lib, name).restype = ctypes.c_int
        lib.svbrdf_host_scenes_max_rows.restype = ctypes.c_int
        lib.svbrdf_scale_inplace.argtypes = [_fp, _fp, ctypes.c_size_t, _fp]
        lib.svbrdf_scale_inplace.restype = ctypes.c_int
        lib.svbrdf_render_fwd_host_scenes.argtypes = [_fp, _fp, ctypes.c_int, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
        lib.svbrdf_render_bwd_host_scenes.argtypes = [_fp, _fp, ctypes.c_int, _fp, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
        for name in ("svbrdf_make_xrow", "svbrdf_render_fwd", "svbrdf_render_bwd", "svbrdf_rendering_loss_fwd_bwd",
                     "svbrdf_render_fwd_host_scenes", "svbrdf_render_bwd_host_scenes"):
            getattr(lib, name).restype = ctypes.c_int
        lib.svbrdf_render_fwd_f64.argtypes = [_fp, _fp, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
        lib.svbrdf_render_bwd_f64.argtypes = [_fp, _fp, _fp, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
        lib.svbrdf_render_fwd_f64.restype = lib.svbrdf_render_bwd_f64.restype = ctypes.c_int
        lib.svbrdf_render_bwd_jvp_f64.argtypes = [_fp] * 7 + [ctypes.c_int] * 4 + [_fp]
        lib.svbrdf_render_bwd_jvp_f64.restype = ctypes.c_int
        lib.svbrdf_render_inputs.argtypes = [_fp, _fp, _fp, ctypes.c_ulonglong, ctypes.c_ulonglong, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
        lib.svbrdf_render_inputs_host_scenes.argtypes = lib.svbrdf_render_inputs.argtypes
        lib.svbrdf_render_inputs.restype = lib.svbrdf_render_inputs_host_scenes.restype = ctypes.c_int
        lib.svbrdf_debug_copy.argtypes = [_fp, _fp, ctypes.c_size_t, _fp]
        lib.svbrdf_debug_copy.restype = ctypes.c_int
        lib.svbrdf_debug_launch_count.argtypes = []
        lib.svbrdf_debug_launch_count.restype = ctypes.c_ulonglong
        v = lib.svbrdf_abi_version()
        if v != ABI_VERSION:
            raise NativeLibraryError("ABI mismatch: library %d, binding %d -- rebuild" % (v, ABI_VERSION))
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = _load().svbrdf_last_error()
        raise NativeLibraryError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def make_xrow_host(W):
    """torch.linspace(-1, 1, W) bit pattern of the reference's CPU path (renderers.py:73)."""
    buf = (ctypes.c_float * W)()
    _check(_load().svbrdf_make_xrow(ctypes.cast(buf, _fp), W), "svbrdf_make_xrow")
    return torch.tensor(list(buf), dtype=torch.float32)


def _require_device_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise NativeLibraryError(
            "%s is on %s: the MI355X engine only computes on a ROCm device (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))


def xrow(device, W):
    key = (device.index, W)
    t = _xrow_cache.get(key)
    if t is None:
        t = make_xrow_host(W).to(device)
        _xrow_cache[key] = t
    return t


def _workspace(device, nbytes):
    # one scratch buffer per (device, stream): calls on different streams never share it
    key = (device.index, _raw_stream(device))
    t = _workspace_cache.get(key)
    if t is None or t.numel() * 8 < nbytes:
        # zero-initialised once; every completed kernel leaves it zeroed (see svbrdf_hip.h)
        t = torch.zeros((max(nbytes, 64) + 7) // 8, dtype=torch.int64, device=device)
        _workspace_cache[key] = t
    return t


def _raw_stream(device):
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


def _stream(device):
    return ctypes.c_void_p(_raw_stream(device))


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (the common case costs ~nothing)"""

    __slots__ = ("ctx",)

    def __init__(self, device):
        self.ctx = None if device.index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _dims(maps, scenes):
    """-> (B, S, H, W, shared): `scenes` is [B,S,9], or [S,9] = the same S scenes for every map (host tables only)"""
    if not isinstance(scenes, torch.Tensor):
        raise TypeError("scenes must be a torch.Tensor")
    if maps.dim() != 4 or maps.shape[1] != 12:
        raise ValueError("maps must be [B,12,H,W], got %s" % (tuple(maps.shape),))
    B, _, H, W = maps.shape
    if H != W:
        raise ValueError("H must equal W (got %dx%d): the reference transposes the x grid, renderers.py:75" % (H, W))
    if scenes.dim() == 2 and not scenes.is_cuda and scenes.shape[1] == 9:
        return B, scenes.shape[0], H, W, True
    if scenes.dim() != 3 or scenes.shape[0] != B or scenes.shape[2] != 9:
        raise ValueError("scenes must be [B,S,9], got %s for B=%d" % (tuple(scenes.shape), B))
    return B, scenes.shape[1], H, W, False


def _scene_table_for_launch(scenes, device):
    """A HOST fp32 table that fits rides in the launch's kernel-argument block (-> host tensor, True); a larger one
    is uploaded, shared rows expanded per map by the caller (-> device tensor, False); a device table passes."""
    if not isinstance(scenes, torch.Tensor):
        raise TypeError("scenes must be a torch.Tensor")
    if scenes.dtype != torch.float32:
        raise TypeError("scenes must be float32 (got %s)" % scenes.dtype)
    if scenes.is_cuda:
        if scenes.device != device:     # the raw pointer would be dereferenced by a kernel running on `device`
            raise ValueError("scenes are on %s, the maps on %s: a device scene table must live with the maps"
                             % (scenes.device, device))
        return (scenes if scenes.is_contiguous() else scenes.contiguous()), False
    rows = scenes.numel() // 9
    if rows <= host_scenes_max_rows():
        return (scenes if scenes.is_contiguous() else scenes.contiguous()), True
    return upload_scene_table(scenes, device), False


def _require_device_float(t, name):
    """float32 (the engine's path) or float64 (the reference's mixed-precision behaviour for double maps)"""
    if isinstance(t, torch.Tensor) and t.dtype == torch.float64 and t.is_cuda:
        return True
    _require_device_f32(t, name)
    return False


def _scene_table_f64(maps, scenes):
    """the float32 DEVICE table [B,S,9] the float64 entry points take (shared rows expanded per map)"""
    B, S, H, W, shared = _dims(maps, scenes)
    if scenes.dtype != torch.float32:
        raise TypeError("scenes must be float32 (got %s): positions and colours are float32 in the reference whatever the "
                        "maps' dtype (torch.Tensor(...), renderers.py:79,91,98)" % scenes.dtype)
    if shared:
        scenes = scenes.unsqueeze(0).expand(B, S, 9)
    if scenes.is_cuda and scenes.device != maps.device:
        raise ValueError("scenes are on %s, the maps on %s" % (scenes.device, maps.device))
    return scenes.to(maps.device).contiguous(), B, S, H, W


def _render_fwd_f64(maps, scenes):
    table, B, S, H, W = _scene_table_f64(maps, scenes)
    out = torch.empty((B, S, 3, H, W), dtype=torch.float64, device=maps.device)
    with _on_device(maps.device):
        _check(_load().svbrdf_render_fwd_f64(maps.data_ptr(), table.data_ptr(), xrow(maps.device, W).data_ptr(),
                                             out.data_ptr(), B, S, H, W, _stream(maps.device)), "svbrdf_render_fwd_f64")
    return out


def _render_bwd_f64(maps, scenes, grad_out):
    table, B, S, H, W = _scene_table_f64(maps, scenes)
    if grad_out.dtype != torch.float64 or not grad_out.is_cuda:
        raise TypeError("grad_out must be a float64 device tensor for float64 maps")
    if grad_out.numel() != B * S * 3 * H * W or grad_out.shape[-2:] != maps.shape[-2:]:
        raise ValueError("grad_out must be [B,S,3,H,W]")
    grad = torch.empty_like(maps)
    with _on_device(maps.device):
        _check(_load().svbrdf_render_bwd_f64(maps.data_ptr(), table.data_ptr(), xrow(maps.device, W).data_ptr(),
                                             grad_out.contiguous().data_ptr(), grad.data_ptr(), B, S, H, W,
                                             _stream(maps.device)), "svbrdf_render_bwd_f64")
    return grad


def render_bwd_jvp_f64(maps, tangent, scenes, grad_out):
    """Second order (svbrdf_render_bwd_jvp_f64): for float64 ``maps`` [B,12,H,W], a direction ``tangent`` of the same
    shape and ``grad_out`` [B,S,3,H,W] -> (d/dmaps <J^T grad_out, tangent> [B,12,H,W],  J tangent [B,S,3,H,W]): the two
    products autograd needs to differentiate through the backward of ``render`` (create_graph=True)."""
    for t, name in ((maps, "maps"), (tangent, "tangent"), (grad_out, "grad_out")):
        if not (isinstance(t, torch.Tensor) and t.dtype == torch.float64 and t.is_cuda):
            raise TypeError("%s must be a float64 tensor on a ROCm device" % name)
    maps, tangent, grad_out = maps.contiguous(), tangent.contiguous(), grad_out.contiguous()
    table, B, S, H, W = _scene_table_f64(maps, scenes)
    if tangent.shape != maps.shape or tangent.device != maps.device:
        raise ValueError("tangent must have the maps' shape and device")
    if grad_out.numel() != B * S * 3 * H * W or grad_out.shape[-2:] != maps.shape[-2:] or grad_out.device != maps.device:
        raise ValueError("grad_out must be [B,S,3,H,W] on the maps' device")
    gm_t = torch.empty_like(maps)
    out_t = torch.empty((B, S, 3, H, W), dtype=torch.float64, device=maps.device)
    with _on_device(maps.device):
        _check(_load().svbrdf_render_bwd_jvp_f64(maps.data_ptr(), tangent.data_ptr(), table.data_ptr(),
                                                 xrow(maps.device, W).data_ptr(), grad_out.data_ptr(), gm_t.data_ptr(),
                                                 out_t.data_ptr(), B, S, H, W, _stream(maps.device)),
               "svbrdf_render_bwd_jvp_f64")
    return gm_t, out_t


def render_fwd(maps, scenes):
    """K1: maps [B,12,H,W]; scenes [B,S,9] on the device, or on the HOST as [B,S,9] / [S,9] (the same S scenes for
    every map): a host table of at most host_scenes_max_rows() rows travels with the launch (one dispatch, no copy
    command).  -> renderings [B,S,3,H,W].  float64 maps take the mixed-precision path of the reference (float32
    geometry, double shading: svbrdf_render_fwd_f64) and return float64."""
    if _require_device_float(maps, "maps"):
        return _render_fwd_f64(maps if maps.is_contiguous() else maps.contiguous(), scenes)
    if not maps.is_contiguous():
        maps = maps.contiguous()
    B, S, H, W, shared = _dims(maps, scenes)
    if shared and S > host_scenes_max_rows():
        scenes, shared = scenes.unsqueeze(0).expand(B, S, 9), False
    table, on_host = _scene_table_for_launch(scenes, maps.device)
    out = torch.empty((B, S, 3, H, W), dtype=torch.float32, device=maps.device)
    lib = _load()
    with _on_device(maps.device):
        if on_host:
            _check(lib.svbrdf_render_fwd_host_scenes(maps.data_ptr(), table.data_ptr(), int(shared),
                                                     xrow(maps.device, W).data_ptr(), out.data_ptr(), B, S, H, W,
                                                     _stream(maps.device)), "svbrdf_render_fwd_host_scenes")
        else:
            _check(lib.svbrdf_render_fwd(maps.data_ptr(), table.data_ptr(), xrow(maps.device, W).data_ptr(),
                                         out.data_ptr(), B, S, H, W, _stream(maps.device)), "svbrdf_render_fwd")
    return out


def device_philox_state(device, generator=None):
    """(seed, offset) for one launch of a counter-based noise kernel, taken from -- and advancing -- torch's generator of
    `device` the way torch's own device random ops do: ``torch.cuda.manual_seed`` therefore controls the sensor noise of
    ``render_inputs`` like it controls ``torch.randn(..., device=...)``.  The kernel uses the offset as the upper counter
    words, so advancing it by one unit (4, the granularity torch requires) gives the next launch a disjoint counter space."""
    gen = generator if generator is not None else torch.cuda.default_generators[
        device.index if device.index is not None else torch.cuda.current_device()]
    seed, offset = int(gen.initial_seed()), int(gen.get_offset())
    gen.set_offset(offset + 4)
    return seed & 0xFFFFFFFFFFFFFFFF, offset & 0xFFFFFFFFFFFFFFFF


def render_inputs(maps, scenes, noise_std=None, seed=0, offset=0):
    """K1 + sensor-noise epilogue (svbrdf_render_inputs*): maps [B,12,H,W] device; scenes [B,S,9] and noise_std [B,S]
    (or None: clamp only) BOTH on the host (at most host_scenes_max_rows() rows: they ride in the launch's argument block)
    or both on the maps' device -> clamp(render + noise_std * N(0,1), 0, 1) [B,S,3,H,W], ONE launch, each photo written
    once.  The normal field is a pure function of (seed, offset, element index): see include/svbrdf_hip.h."""
    _require_device_f32(maps, "maps")
    if not maps.is_contiguous():
        maps = maps.contiguous()
    B, S, H, W, shared = _dims(maps, scenes)
    if shared:
        raise ValueError("render_inputs needs one scene row per photo: scenes must be [B,S,9]")
    if scenes.dtype != torch.float32:
        raise TypeError("scenes must be float32 (got %s)" % scenes.dtype)
    if noise_std is not None:
        if not isinstance(noise_std, torch.Tensor) or noise_std.dtype != torch.float32 or noise_std.numel() != B * S:
            raise ValueError("noise_std must be a float32 tensor of B*S = %d levels" % (B * S))
        if noise_std.is_cuda != scenes.is_cuda:
            raise ValueError("scenes and noise_std must both be on the host or both on the maps' device")
    on_host = not scenes.is_cuda
    if on_host and B * S > host_scenes_max_rows():
        scenes = upload_scene_table(scenes, maps.device)
        noise_std = upload_scene_table(noise_std.reshape(-1), maps.device) if noise_std is not None else None
        on_host = False
    if not on_host and (scenes.device != maps.device or (noise_std is not None and noise_std.device != maps.device)):
        raise ValueError("device scene / noise tables must live with the maps")
    scenes = scenes.contiguous()
    sig = noise_std.contiguous() if noise_std is not None else None
    out = torch.empty((B, S, 3, H, W), dtype=torch.float32, device=maps.device)
    lib = _load()
    fn = lib.svbrdf_render_inputs_host_scenes if on_host else lib.svbrdf_render_inputs
    with _on_device(maps.device):
        _check(fn(maps.data_ptr(), scenes.data_ptr(), sig.data_ptr() if sig is not None else None,
                  ctypes.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF), ctypes.c_ulonglong(int(offset) & 0xFFFFFFFFFFFFFFFF),
                  xrow(maps.device, W).data_ptr(), out.data_ptr(), B, S, H, W, _stream(maps.device)),
               "svbrdf_render_inputs_host_scenes" if on_host else "svbrdf_render_inputs")
    return out


def debug_copy(dst, src):
    """svbrdf_debug_copy: dst[:] = src as a streaming float4 copy kernel on the current stream (measurement aid: the copy
    bandwidth of this box, bench.py / tests/test_gpu_perf_guard.py)"""
    _require_device_f32(dst, "dst")
    _require_device_f32(src, "src")
    if dst.numel() != src.numel() or not dst.is_contiguous() or not src.is_contiguous() or dst.device != src.device:
        raise ValueError("debug_copy needs two contiguous tensors of one size on one device")
    with _on_device(dst.device):
        _check(_load().svbrdf_debug_copy(dst.data_ptr(), src.data_ptr(), dst.numel(), _stream(dst.device)), "svbrdf_debug_copy")
    return dst


def launch_count():
    """kernels enqueued by libsvbrdf_hip.so in this process so far (svbrdf_debug_launch_count)"""
    return int(_load().svbrdf_debug_launch_count())


def render_bwd(maps, scenes, grad_out):
    """K2: adjoint of render_fwd (same `scenes` forms) -> grad_maps [B,12,H,W]."""
    if _require_device_float(maps, "maps"):
        return _render_bwd_f64(maps if maps.is_contiguous() else maps.contiguous(), scenes, grad_out)
    _require_device_f32(grad_out, "grad_out")
    if not maps.is_contiguous():
        maps = maps.contiguous()
    if not grad_out.is_contiguous():
        grad_out = grad_out.contiguous()
    B, S, H, W, shared = _dims(maps, scenes)
    if grad_out.numel() != B * S * 3 * H * W or grad_out.shape[-2:] != maps.shape[-2:]:
        raise ValueError("grad_out must be [B,S,3,H,W]")
    if shared and S > host_scenes_max_rows():
        scenes, shared = scenes.unsqueeze(0).expand(B, S, 9), False
    table, on_host = _scene_table_for_launch(scenes, maps.device)
    grad = torch.empty_like(maps)
    lib = _load()
    with _on_device(maps.device):
        if on_host:
            _check(lib.svbrdf_render_bwd_host_scenes(maps.data_ptr(), table.data_ptr(), int(shared),
                                                     xrow(maps.device, W).data_ptr(), grad_out.data_ptr(), grad.data_ptr(),
                                                     B, S, H, W, _stream(maps.device)), "svbrdf_render_bwd_host_scenes")
        else:
            _check(lib.svbrdf_render_bwd(maps.data_ptr(), table.data_ptr(), xrow(maps.device, W).data_ptr(),
                                         grad_out.data_ptr(), grad.data_ptr(), B, S, H, W, _stream(maps.device)),
                   "svbrdf_render_bwd")
    return grad


def _ragged_offsets(counts, B, R, device):
    counts = [int(c) for c in counts]
    if len(counts) != B or any(c < 0 for c in counts) or sum(counts) != R:
        raise ValueError("counts must hold one non-negative render count per map and sum to the number of scenes")
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    return torch.tensor(off, dtype=torch.int32).to(device)


def _ragged_binding():
    lib = _load()
    lib.svbrdf_render_fwd_ragged.argtypes = [_fp, _fp, _fp, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
    lib.svbrdf_render_bwd_ragged.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp] + [ctypes.c_int] * 4 + [_fp]
    lib.svbrdf_render_fwd_ragged.restype = lib.svbrdf_render_bwd_ragged.restype = ctypes.c_int
    return lib


def render_fwd_ragged(maps, scenes, counts):
    """K1, ragged: maps [B,12,H,W], scenes [R,9] grouped by map, counts[b] renders for map b -> [R,3,H,W]."""
    _require_device_f32(maps, "maps")
    _require_device_f32(scenes, "scenes")
    maps, scenes = maps.contiguous(), scenes.contiguous()
    if maps.dim() != 4 or maps.shape[1] != 12 or maps.shape[2] != maps.shape[3] or scenes.dim() != 2 or scenes.shape[1] != 9:
        raise ValueError("maps must be [B,12,H,H] and scenes [R,9]")
    B, _, H, W = maps.shape
    R = scenes.shape[0]
    off = _ragged_offsets(counts, B, R, maps.device)
    out = torch.empty((R, 3, H, W), dtype=torch.float32, device=maps.device)
    with _on_device(maps.device):
        _check(_ragged_binding().svbrdf_render_fwd_ragged(maps.data_ptr(), scenes.data_ptr(), off.data_ptr(),
                                                          xrow(maps.device, W).data_ptr(), out.data_ptr(), B, R, H, W,
                                                          _stream(maps.device)), "svbrdf_render_fwd_ragged")
    return out


def render_bwd_ragged(maps, scenes, counts, grad_out):
    """K2, ragged: adjoint of render_fwd_ragged -> grad_maps [B,12,H,W] (zeros for a map without renders)."""
    for t, name in ((maps, "maps"), (scenes, "scenes"), (grad_out, "grad_out")):
        _require_device_f32(t, name)
    maps, scenes, grad_out = maps.contiguous(), scenes.contiguous(), grad_out.contiguous()
    B, _, H, W = maps.shape
    R = scenes.shape[0]
    if tuple(grad_out.shape) != (R, 3, H, W):
        raise ValueError("grad_out must be [R,3,H,W]")
    off = _ragged_offsets(counts, B, R, maps.device)
    grad = torch.empty_like(maps)
    with _on_device(maps.device):
        _check(_ragged_binding().svbrdf_render_bwd_ragged(maps.data_ptr(), scenes.data_ptr(), off.data_ptr(),
                                                          xrow(maps.device, W).data_ptr(), grad_out.data_ptr(),
                                                          grad.data_ptr(), B, R, H, W, _stream(maps.device)),
               "svbrdf_render_bwd_ragged")
    return grad


def rendering_loss(input, target, scenes, eps=0.1, want_grad=True, l1_weight=0.0, eps_l1=0.01, head=False):
    """K3: fused rendering loss (+ d loss/d input); with l1_weight != 0 the SVBRDF L1 loss is folded
    in (MixedLoss); with head=True `input` is the generator's [B,9,H,W] post-tanh output and the
    network head is decoded in the kernel.  Returns (loss [1] device tensor, grad or None)."""
    _require_device_f32(input, "input")
    _require_device_f32(target, "target")
    # a HOST table of at most host_scenes_max_rows() rows rides in the kernel-argument block (no upload)
    host_scenes = isinstance(scenes, torch.Tensor) and not scenes.is_cuda
    if host_scenes:
        if scenes.dtype != torch.float32:
            raise TypeError("scenes must be float32 (got %s)" % scenes.dtype)
        if scenes.dim() == 3 and scenes.shape[0] * scenes.shape[1] > host_scenes_max_rows():
            scenes, host_scenes = upload_scene_table(scenes, input.device), False
    else:
        _require_device_f32(scenes, "scenes")
    if head:
        if input.dim() != 4 or input.shape[1] != 9 or (input.shape[0],) + tuple(input.shape[2:]) != \
                (target.shape[0],) + tuple(target.shape[2:]):
            raise ValueError("head=True needs input [B,9,H,W] and target [B,12,H,W]")
    elif input.shape != target.shape:
        raise ValueError("input and target shapes differ: %s vs %s" % (tuple(input.shape), tuple(target.shape)))
    if target.device != input.device or (not host_scenes and scenes.device != input.device):
        raise ValueError("input, target and scenes must be on the same device")
    input, target, scenes = input.contiguous(), target.contiguous(), scenes.contiguous()
    B, S, H, W, shared = _dims(target, scenes)
    if shared:
        raise ValueError("the loss needs one scene table per batch item: scenes must be [B,S,9]")
    lib = _load()
    nbytes = lib.svbrdf_rendering_loss_workspace_bytes(B, S, H, W)
    ws = _workspace(input.device, nbytes)
    loss = torch.empty(1, dtype=torch.float32, device=input.device)
    grad = torch.empty_like(input) if want_grad else None
    xr = xrow(input.device, W)
    hook = _launch_hook
    with _on_device(input.device):
        if hook is not None:
            hook("begin")
        if host_scenes:
            entry = "svbrdf_head_loss_fwd_bwd_host_scenes" if head else "svbrdf_mixed_loss_fwd_bwd_host_scenes"
            rc = getattr(lib, entry)(
                input.data_ptr(), target.data_ptr(), scenes.data_ptr(), xr.data_ptr(),
                ctypes.c_float(eps), ctypes.c_float(l1_weight), ctypes.c_float(eps_l1), loss.data_ptr(),
                grad.data_ptr() if want_grad else None, ws.data_ptr(), ws.numel() * 8, B, S, H, W,
                _stream(input.device))
        elif head:
            entry = "svbrdf_head_loss_fwd_bwd"
            rc = lib.svbrdf_head_loss_fwd_bwd(
                input.data_ptr(), target.data_ptr(), scenes.data_ptr(), xr.data_ptr(),
                ctypes.c_float(eps), ctypes.c_float(l1_weight), ctypes.c_float(eps_l1), loss.data_ptr(),
                grad.data_ptr() if want_grad else None, ws.data_ptr(), ws.numel() * 8, B, S, H, W,
                _stream(input.device))
        elif l1_weight != 0.0:
            entry = "svbrdf_mixed_loss_fwd_bwd"
            rc = lib.svbrdf_mixed_loss_fwd_bwd(
                input.data_ptr(), target.data_ptr(), scenes.data_ptr(), xr.data_ptr(),
                ctypes.c_float(eps), ctypes.c_float(l1_weight), ctypes.c_float(eps_l1), loss.data_ptr(),
                grad.data_ptr() if want_grad else None, ws.data_ptr(), ws.numel() * 8, B, S, H, W,
                _stream(input.device))
        else:
            entry = "svbrdf_rendering_loss_fwd_bwd"
            rc = lib.svbrdf_rendering_loss_fwd_bwd(
                input.data_ptr(), target.data_ptr(), scenes.data_ptr(), xr.data_ptr(),
                ctypes.c_float(eps), loss.data_ptr(), grad.data_ptr() if want_grad else None,
                ws.data_ptr(), ws.numel() * 8, B, S, H, W, _stream(input.device))
        if hook is not None:
            hook("end")
    if rc != 0:
        # a failed launch may leave partial sums / arrival counts behind: the scratch contract ("every completed
        # call leaves it zeroed") only covers completed calls, so restore it before reporting the error
        ws.zero_()
    _check(rc, entry)
    return loss, grad


def mix_materials(svbrdf0, svbrdf1, alpha):
    """K4: out[b] = mix(svbrdf0[b], svbrdf1[b], alpha[b]) (dataset.py:142-160) for [B,12,H,W] device tensors and a
    [B] device tensor of blend weights."""
    for t, name in ((svbrdf0, "svbrdf0"), (svbrdf1, "svbrdf1"), (alpha, "alpha")):
        _require_device_f32(t, name)
    if svbrdf0.dim() != 4 or svbrdf0.shape[1] != 12 or svbrdf0.shape != svbrdf1.shape:
        raise ValueError("svbrdf0 and svbrdf1 must both be [B,12,H,W]")
    B, _, H, W = svbrdf0.shape
    if alpha.numel() != B:
        raise ValueError("alpha must hold one weight per batch item")
    if svbrdf1.device != svbrdf0.device or alpha.device != svbrdf0.device:
        raise ValueError("svbrdf0, svbrdf1 and alpha must be on the same device")
    a, b, w = svbrdf0.contiguous(), svbrdf1.contiguous(), alpha.contiguous().view(-1)
    out = torch.empty_like(a)
    lib = _load()
    lib.svbrdf_mix_materials.argtypes = [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]
    lib.svbrdf_mix_materials.restype = ctypes.c_int
    with _on_device(a.device):
        _check(lib.svbrdf_mix_materials(a.data_ptr(), b.data_ptr(), w.data_ptr(), out.data_ptr(), B, H, W,
                                        _stream(a.device)), "svbrdf_mix_materials")
    return out


def clock_probe(out, ticks=300000, stream=None):
    """Measurement aid (svbrdf_debug_clock_probe): one wave spins for `ticks` ticks of the 100 MHz counter on
    `stream` (a torch.cuda.Stream; default: the current one) and writes shader cycles / ticks into `out`
    (device int64[2]).  cycles / ticks * 0.1 = shader clock in GHz under whatever else is running."""
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.int64 and out.numel() >= 2):
        raise TypeError("out must be a device int64 tensor of at least 2 elements")
    lib = _load()
    lib.svbrdf_debug_clock_probe.argtypes = [_fp, ctypes.c_ulonglong, _fp]
    lib.svbrdf_debug_clock_probe.restype = ctypes.c_int
    raw = stream.cuda_stream if stream is not None else _raw_stream(out.device)
    with _on_device(out.device):
        _check(lib.svbrdf_debug_clock_probe(out.data_ptr(), int(ticks), ctypes.c_void_p(raw)), "svbrdf_debug_clock_probe")
    return out


_host_rows = None


def host_scenes_max_rows():
    """largest B*S the *_host_scenes entry points take (the table rides in the launch's kernel-argument block)"""
    global _host_rows
    if _host_rows is None:
        _host_rows = int(_load().svbrdf_host_scenes_max_rows())
    return _host_rows


def scale_inplace_(data, scale):
    """data *= scale (a one-element device tensor) without a host sync; no-op kernel when scale == 1."""
    _require_device_f32(data, "data")
    _require_device_f32(scale, "scale")
    if not data.is_contiguous() or scale.numel() != 1:
        raise ValueError("scale_inplace_ needs a contiguous tensor and a one-element scale")
    with _on_device(data.device):
        _check(_load().svbrdf_scale_inplace(data.data_ptr(), scale.data_ptr(), data.numel(), _stream(data.device)),
               "svbrdf_scale_inplace")
    return data


class _PinnedRing:
    """Truly asynchronous upload of the small per-call scene table.

    A pageable-memory ``hipMemcpyAsync`` stages through the runtime and makes the host wait for
    the stream (measured: the step time was host + GPU instead of max(host, GPU)).  The table
    is therefore copied into one of a few pinned slots and uploaded from there; a slot is
    reused only after the event recorded behind its upload has completed."""

    def __init__(self, depth=8):
        self.depth, self.slots, self.events, self.next = depth, [None] * depth, [None] * depth, 0

    def upload(self, host, device):
        i = self.next
        self.next = (i + 1) % self.depth
        if self.events[i] is not None:
            self.events[i].synchronize()            # normally long done: depth steps ago
        slot = self.slots[i]
        if slot is None or slot.numel() < host.numel():
            slot = torch.empty(max(host.numel(), 1024), dtype=torch.float32, pin_memory=True)
            self.slots[i] = slot
        view = slot[:host.numel()].view(host.shape)
        view.copy_(host)
        dev = view.to(device, non_blocking=True)
        ev = self.events[i] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[i] = ev
        return dev


_rings = {}


def upload_scene_table(host_table, device):
    """[B,S,9] fp32 host tensor -> device tensor, without stalling the host on the stream."""
    if host_table.dtype != torch.float32:
        raise TypeError("scenes must be float32 (got %s)" % host_table.dtype)
    ring = _rings.get(device.index)
    if ring is None:
        ring = _rings[device.index] = _PinnedRing()
    return ring.upload(host_table.contiguous(), device)
