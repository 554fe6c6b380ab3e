"""``LocalRenderer`` -- the in-network point-light Cook-Torrance/GGX renderer, MI355X-native.

Keeps the reference's plugin interface (development/multiImage_pytorch/renderers.py:14,
:67): ``LocalRenderer().render(scene, svbrdf) -> radiance`` with
  * ``svbrdf`` fp32 ``[12,H,W]`` -> ``[1,3,H,W]``, or ``[B,12,H,W]`` -> ``[B,3,H,W]`` with the
    ONE scene applied to every batch item (renderers.py:98 makes the light colour 4-D),
  * ``scene`` any object with ``.camera.pos``, ``.light.pos``, ``.light.color``
    (3-sequences, ndarrays or tensors -- dataset.py:210 passes tensors),
  * differentiable w.r.t. ``svbrdf``.
  * a HOST ``svbrdf`` (the reference's dataloader renders missing input photos in the main process with CPU tensors,
    dataset.py:94-98 -> :206-212) is staged through pinned memory to the current ROCm device, rendered by the same
    kernel K1 and returned as a CPU tensor (forward only; see ``_HostStaging``).  Without a usable ROCm device the
    call raises ``NativeLibraryError`` -- there is no CPU implementation.
The arithmetic (renderers.py:8-104) runs in the hand-written HIP kernels K1 (forward) and
K2 (analytic backward, forward recomputed in registers) of csrc/svbrdf_kernels.hip.

Extension (not in the reference): ``render_many(scene_table, svbrdf)`` renders S scenes per
map in one launch, reading the maps once.
"""
import threading

import torch

from . import _hostext, _native, environment


class _RenderFunction(torch.autograd.Function):
    """maps [B,12,H,W]; scenes (no grad) [B,S,9] on the device, or on the host as [B,S,9] / [S,9] (shared by all
    maps): a small host table rides in the launch's argument block (_native.render_fwd) -> [B,S,3,H,W]"""

    @staticmethod
    def forward(ctx, maps, scenes):
        if not maps.is_contiguous():
            maps = maps.contiguous()
        if scenes.is_cuda:
            ctx.save_for_backward(maps, scenes)                    # version-checked like any saved tensor
            ctx.host_scenes = None
        else:
            ctx.save_for_backward(maps)
            ctx.host_scenes = scenes.detach().clone()              # a few rows: the caller may reuse its buffer
        return _native.render_fwd(maps, scenes)

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.needs_input_grad[0]:
            return None, None
        saved = ctx.saved_tensors
        scenes = ctx.host_scenes if ctx.host_scenes is not None else saved[1]
        if torch.is_grad_enabled():
            # backward(create_graph=True): the gradient must itself be differentiable, as it is through the reference's
            # op-by-op render (renderers.py:67-104)
            return differentiable_render_backward(saved[0], scenes, grad_out), None
        return _native.render_bwd(saved[0], scenes, grad_out), None


class _RenderBackwardF64(torch.autograd.Function):
    """``J(maps)^T grad_out`` in float64 (K2's float64 form) as a node that can be differentiated once more: its backward is
    the dual-number kernel ``svbrdf_render_bwd_jvp_f64``.  maps [B,12,H,W], grad_out [B,S,3,H,W] double; scenes [B,S,9]
    float32 on the device."""

    @staticmethod
    def forward(ctx, maps, grad_out, scenes):
        ctx.save_for_backward(maps, grad_out, scenes)
        return _native.render_bwd(maps, scenes, grad_out)

    @staticmethod
    def backward(ctx, tangent):
        maps, grad_out, scenes = ctx.saved_tensors
        third = torch.is_grad_enabled()                   # create_graph=True on the SECOND-order call
        with torch.no_grad():
            grad_maps_t, out_t = _native.render_bwd_jvp_f64(maps, tangent, scenes, grad_out)
        if third:
            # there is no third-order kernel: an error when someone differentiates these, never silent constants
            # (torch's once_differentiable only guards against a tangent that requires grad, not against the saved maps)
            grad_maps_t = _NoThirdOrder.apply(grad_maps_t, maps, grad_out, tangent)
            out_t = _NoThirdOrder.apply(out_t, maps, grad_out, tangent)
        return (grad_maps_t if ctx.needs_input_grad[0] else None), (out_t if ctx.needs_input_grad[1] else None), None


class _NoThirdOrder(torch.autograd.Function):
    """identity whose derivative is an error: marks the second-order results as depending on the maps, the cotangent and
    the direction without offering a derivative"""

    @staticmethod
    def forward(ctx, value, *depends_on):
        return value.view_as(value)

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError("third-order derivatives of LocalRenderer.render are not implemented by the MI355X engine "
                           "(first order: K2 / K3; second order: svbrdf_render_bwd_jvp_f64)")


def differentiable_render_backward(maps, scenes, grad_out):
    """The gradient of ``render`` w.r.t. ``maps`` for upstream ``grad_out``, attached to the autograd graph of both (what
    ``create_graph=True`` asks for).  Computed in float64 whatever the maps' dtype (the float32 kernels K1/K2/K3 have no
    second-order companion; float32 maps are promoted exactly) and returned in the maps' dtype.  Also the entry point the
    native host extension calls back into (csrc/host_ext.cpp) when its nodes run under create_graph=True."""
    table = _native._scene_table_f64(maps, scenes)[0]
    grad = _RenderBackwardF64.apply(maps.to(torch.float64), grad_out.reshape(table.shape[0], table.shape[1], 3, *maps.shape[-2:])
                                    .to(torch.float64), table)
    return grad.to(maps.dtype)


class _HostStaging:
    """The reference's SECOND caller of ``render`` hands it host tensors: ``SvbrdfDataset.render_inputs``
    (dataset.py:206-212) runs in the main process (main.py:63: ``num_workers=0``), renders one ``[1,12,H,W]`` CPU map per
    missing input photo and ``torch.cat``s the CPU result with the photos it read (dataset.py:98).  This class serves that
    call shape on the GPU: maps -> pinned slot -> device (async copy) -> K1 with the scene row by value -> pinned slot
    (async copy) -> a fresh pageable CPU tensor, all on the current stream of the current device, one event wait at the
    end.  Slots are cached per (device, shape): a dataloader calls with one shape for its whole life.  Forward only."""

    _lock = threading.Lock()
    _slots = {}

    @classmethod
    def usable_device(cls):
        """the device a host tensor is rendered on, or a NativeLibraryError saying why there is none"""
        bad_fork = getattr(torch.cuda, "_is_in_bad_fork", None)
        if bad_fork is not None and bad_fork():
            raise _native.NativeLibraryError(
                "LocalRenderer.render got a CPU tensor in a forked worker process whose parent had already initialised the "
                "ROCm runtime: a forked child cannot use the GPU, and this engine has no CPU implementation.  Use "
                "num_workers=0 (the reference's main.py:63), a 'spawn' DataLoader context, synthesis.render_inputs on the "
                "device, or keep the reference's own renderer for the dataloader: svbrdf_estimation_amd.install("
                "patch_renderer=False)")
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError(
                "LocalRenderer.render got a CPU tensor and no ROCm device is available: the MI355X engine only computes on "
                "the GPU (no CPU fallback).  On a machine without a GPU keep the reference's own renderer: "
                "svbrdf_estimation_amd.install(patch_renderer=False)")
        return torch.device("cuda", torch.cuda.current_device())

    @classmethod
    def render(cls, maps, row):
        """maps: host [B,12,H,W] (any strides); row: host [1,9] -> host [B,3,H,W]"""
        dev = cls.usable_device()
        if maps.dtype != torch.float32:
            # double maps: the reference's mixed-precision path, not worth pinned slots of its own
            with torch.no_grad():
                return _native.render_fwd(maps.detach().to(dev), row).view(maps.shape[0], 3, *maps.shape[-2:]).cpu()
        key = (dev.index, tuple(maps.shape))
        with cls._lock:
            slot = cls._slots.get(key)
            if slot is None:
                B, _, H, W = maps.shape
                slot = (torch.empty(maps.shape, dtype=torch.float32, pin_memory=True),
                        torch.empty(maps.shape, dtype=torch.float32, device=dev),
                        torch.empty((B, 3, H, W), dtype=torch.float32, pin_memory=True), torch.cuda.Event())
                if len(cls._slots) >= 8:            # a new shape after eight: drop the oldest set of buffers
                    cls._slots.pop(next(iter(cls._slots)))
                cls._slots[key] = slot
            pin_in, dev_in, pin_out, done = slot
            with torch.no_grad():
                pin_in.copy_(maps)                                  # host copy (gathers a strided / expanded view)
                dev_in.copy_(pin_in, non_blocking=True)             # H2D from pinned memory: asynchronous
                out = _native.render_fwd(dev_in, row)               # K1, scene row by value: [B,1,3,H,W]
                pin_out.copy_(out.view(pin_out.shape), non_blocking=True)
                done.record(torch.cuda.current_stream(dev))
                done.synchronize()
                return pin_out.clone()                              # pageable, the caller's to keep (dataset.py:98 cats it)


class LocalRenderer:
    """Drop-in for renderers.LocalRenderer (no constructor arguments)."""

    def render(self, scene, svbrdf):
        if not isinstance(svbrdf, torch.Tensor):
            raise TypeError("svbrdf must be a torch.Tensor")
        if svbrdf.dim() == 3:
            maps = svbrdf.unsqueeze(0)
        elif svbrdf.dim() == 4:
            maps = svbrdf
        else:
            raise ValueError("svbrdf must be [12,H,W] or [B,12,H,W], got %s" % (tuple(svbrdf.shape),))
        if maps.shape[1] != 12 or maps.shape[-1] != maps.shape[-2]:
            raise ValueError("svbrdf must have 12 channels and H == W (the reference transposes the x grid, "
                             "renderers.py:75), got %s" % (tuple(svbrdf.shape),))
        # the scene's nine floats stay on the host and travel with the launch: one dispatch per call, no H2D copy
        # (the reference uploads camera, light and colour with three synchronous copies, renderers.py:79,91,98)
        ext = _hostext.module() if maps.is_cuda else None
        if ext is not None and maps.dtype == torch.float32 and maps.device.index == torch.cuda.current_device():
            # native host path (csrc/host_ext.cpp): same two kernels through the same C ABI, C++ autograd node; the scene goes
            # over as nine python floats (rounded to float32 there like torch.Tensor(...) rounds them)
            out = ext.render_one_scene(maps, environment.scene_floats(scene), _native._raw_stream(maps.device))
            return out.view(maps.shape[0], 3, maps.shape[-2], maps.shape[-1])
        row = environment.scene_to_row(scene).view(1, 9)           # one scene, shared by every map of the batch
        if not maps.is_cuda:
            if maps.requires_grad and torch.is_grad_enabled():
                raise _native.NativeLibraryError(
                    "LocalRenderer.render got a CPU tensor that requires grad: host tensors are rendered forward-only (the "
                    "dataloader's call, dataset.py:206-212); move the maps to the ROCm device to differentiate through render")
            return _HostStaging.render(maps, row)
        out = _RenderFunction.apply(maps, row)                     # ctypes path; raises on non-ROCm tensors
        return out.view(maps.shape[0], 3, maps.shape[-2], maps.shape[-1])

    def render_many(self, scene_table, svbrdf):
        """scene_table [S,9] (shared by all maps) or [B,S,9]; svbrdf [B,12,H,W] -> [B,S,3,H,W]."""
        if svbrdf.dim() != 4:
            raise ValueError("render_many expects svbrdf [B,12,H,W]")
        table = torch.as_tensor(scene_table, dtype=torch.float32)
        if table.is_cuda and table.dim() == 2:
            table = table.unsqueeze(0).expand(svbrdf.shape[0], -1, -1)
        return _RenderFunction.apply(svbrdf, table)
