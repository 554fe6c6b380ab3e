"""``LocalRenderer`` -- the in-network point-light Cook-Torrance/GGX renderer, MI355X-native.

Keeps the reference's plugin interface (development/multiImage_pytorch/renderers.py:14,
:67): ``LocalRenderer().render(scene, svbrdf) -> radiance`` with
  * ``svbrdf`` fp32 ``[12,H,W]`` -> ``[1,3,H,W]``, or ``[B,12,H,W]`` -> ``[B,3,H,W]`` with the
    ONE scene applied to every batch item (renderers.py:98 makes the light colour 4-D),
  * ``scene`` any object with ``.camera.pos``, ``.light.pos``, ``.light.color``
    (3-sequences, ndarrays or tensors -- dataset.py:210 passes tensors),
  * differentiable w.r.t. ``svbrdf``.
The arithmetic (renderers.py:8-104) runs in the hand-written HIP kernels K1 (forward) and
K2 (analytic backward, forward recomputed in registers) of csrc/svbrdf_kernels.hip.

Extension (not in the reference): ``render_many(scene_table, svbrdf)`` renders S scenes per
map in one launch, reading the maps once.
"""
import torch

from . import _native, environment


class _RenderFunction(torch.autograd.Function):
    """maps [B,12,H,W], scenes [B,S,9] (device, no grad) -> [B,S,3,H,W]"""

    @staticmethod
    def forward(ctx, maps, scenes):
        maps = maps.contiguous()
        ctx.save_for_backward(maps, scenes)
        return _native.render_fwd(maps, scenes)

    @staticmethod
    def backward(ctx, grad_out):
        maps, scenes = ctx.saved_tensors
        grad_maps = _native.render_bwd(maps, scenes, grad_out.contiguous()) if ctx.needs_input_grad[0] else None
        return grad_maps, None


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
        B = maps.shape[0]
        row = environment.scene_to_row(scene)                      # host, 9 floats
        table = row.view(1, 1, 9).expand(B, 1, 9).contiguous()
        table = table.to(maps.device, non_blocking=True) if maps.is_cuda else table
        out = _RenderFunction.apply(maps, table)                   # raises on non-ROCm tensors
        return out.view(B, 3, maps.shape[-2], maps.shape[-1])

    def render_many(self, scene_table, svbrdf):
        """scene_table [S,9] (shared by all maps) or [B,S,9]; svbrdf [B,12,H,W] -> [B,S,3,H,W]."""
        if svbrdf.dim() != 4:
            raise ValueError("render_many expects svbrdf [B,12,H,W]")
        table = torch.as_tensor(scene_table, dtype=torch.float32)
        if table.dim() == 2:
            table = table.unsqueeze(0).expand(svbrdf.shape[0], -1, -1)
        table = table.contiguous().to(svbrdf.device, non_blocking=True)
        return _RenderFunction.apply(svbrdf, table)
