"""On-GPU synthesis of the network's input photographs.

The reference renders missing input photos inside its CPU dataloader, one 256x256 image at a
time with the eager ``LocalRenderer`` (development/multiImage_pytorch/dataset.py:94-98 ->
``render_inputs`` :162-221): fronto-parallel first view, cosine-hemisphere views after it,
optional light-power / white-balance / view-distance augmentation, Gaussian sensor noise,
clamp to [0,1].  Here the scenes are drawn on the host with the reference's RNG call sequence
and all ``count`` photos of all batch items are rendered by ONE launch of the forward kernel
K1, reading each SVBRDF once.

Noise: the reference draws the per-pixel noise from the CPU generator.  ``noise="cpu"`` does the
same (bit-identical noise field, costs a 3*H*W draw + upload per image).  ``noise="device"``
(default) fuses noise and clamp into K1's store (``svbrdf_render_inputs``): the levels are drawn on
the host, the field is counter-based (Philox4x32-10 keyed by torch's device generator), the whole
call is ONE kernel launch that writes every photo once -- statistically identical, different
numbers.  ``noise=None``: the same launch without the field (clamp only).
"""
import math

import numpy as np
import torch

from . import _hostext, _native, utils

MIN_EPS, MAX_EPS = 0.001, 0.02           # dataset.py:164-165
FIXED_LIGHT_DISTANCE = 2.197             # dataset.py:166
FIXED_VIEW_DISTANCE = 2.75               # dataset.py:167


def input_scene_table(count, use_augmentation=True):
    """[count,9] scenes of ONE sample, RNG call order of dataset.py:172-204."""
    light = torch.cat((torch.empty(2).uniform_(-0.75, 0.75), torch.ones(1) * FIXED_LIGHT_DISTANCE)).unsqueeze(0)
    if count > 1:
        hemi = utils.generate_normalized_random_direction(count - 1, min_eps=MIN_EPS, max_eps=MAX_EPS)
        light = torch.cat((light, hemi * FIXED_LIGHT_DISTANCE), dim=0)
    colors = torch.tensor([30.0]).unsqueeze(-1)
    if use_augmentation:
        std = torch.exp(torch.empty(1).normal_(mean=-2.0, std=0.5)).numpy()[0]
        colors = torch.abs(torch.empty(count).normal_(mean=20.0, std=std)).unsqueeze(-1)
    colors = colors.expand(count, 3)
    if use_augmentation:
        colors = colors * torch.abs(torch.empty(count, 3).normal_(mean=1.0, std=0.03))   # white balance
        view_distance = torch.empty(count).uniform_(0.25, 2.75)
    else:
        view_distance = torch.ones(count) * FIXED_VIEW_DISTANCE
    view = torch.cat((torch.empty(2).uniform_(-0.25, 0.25), view_distance[:1])).unsqueeze(0)
    if count > 1:
        hemi = utils.generate_normalized_random_direction(count - 1, min_eps=MIN_EPS, max_eps=MAX_EPS)
        view = torch.cat((view, hemi * view_distance[1:].unsqueeze(-1)), dim=0)
    return torch.cat((view, light, colors), dim=-1).contiguous()


def noise_std():
    """per-image noise level, dataset.py:215"""
    return torch.exp(torch.empty(1).normal_(mean=np.log(0.005), std=0.3)).numpy()[0]


def noise_levels(n):
    """the levels of n photos in ONE host draw, same distribution as n calls of noise_std() (dataset.py:215)"""
    return torch.exp(torch.empty(n).normal_(mean=math.log(0.005), std=0.3))


def render_inputs(svbrdf, count, use_augmentation=True, noise="device", generator=None):
    """svbrdf [12,H,W] (one sample, like the reference) or [B,12,H,W] on a ROCm device ->
    [count,3,H,W] / [B,count,3,H,W] linear-RGB input photos in [0,1].  `generator`: the torch device generator the
    ``noise="device"`` field is keyed by (default: the device's default generator, i.e. ``torch.cuda.manual_seed``)."""
    single = svbrdf.dim() == 3
    maps = svbrdf.unsqueeze(0) if single else svbrdf
    if maps.dim() != 4 or maps.shape[1] != 12:
        raise ValueError("svbrdf must be [12,H,W] or [B,12,H,W]")
    B, _, H, W = maps.shape
    if noise not in (None, "cpu", "device"):
        raise ValueError("noise must be None, 'cpu' or 'device'")
    ext = _hostext.module()          # native sampler (csrc/host_ext.cpp): the same draws in the same order, bit for bit
    if noise == "cpu":
        tables, fields = [], []
        for _ in range(B):
            tables.append(ext.sample_input_scene_table(1, count, bool(use_augmentation))[0] if ext is not None
                          else input_scene_table(count, use_augmentation))
            per_image = []          # reference order: after a sample's scenes, per image: level, then the field
            for _i in range(count):
                std = noise_std()
                per_image.append(torch.zeros(1, 3, H, W).normal_(mean=0.0, std=std))
            fields.append(torch.cat(per_image, dim=0))
        table = torch.stack(tables, dim=0)
    elif ext is not None:
        table = ext.sample_input_scene_table(B, count, bool(use_augmentation))       # sample after sample, one call
    else:
        table = torch.stack([input_scene_table(count, use_augmentation) for _ in range(B)], dim=0)
    if noise == "cpu":
        # K1: [B,count,3,H,W]; the host table travels with the launch when it fits the argument block (pinned ring otherwise)
        out = _native.render_fwd(maps.detach(), table)
        out = out + torch.stack(fields, dim=0).to(maps.device, non_blocking=True)
        out = out.clamp_(0.0, 1.0)
    else:
        # ONE launch: render + sigma * N(0,1) + clamp in K1's epilogue; levels of all B*count photos in one host draw
        levels = noise_levels(B * count).view(B, count) if noise == "device" else None
        seed, offset = _native.device_philox_state(maps.device, generator) if noise == "device" else (0, 0)
        out = _native.render_inputs(maps.detach(), table, levels, seed, offset)
    return out[0] if single else out


def draw_mix_alpha():
    """the blend weight of one mixed sample, dataset.py:144: U(0.1, 0.9) from torch's global CPU generator"""
    return torch.empty(1).uniform_(0.1, 0.9)


def mix_materials(svbrdf_0, svbrdf_1, alpha=None):
    """``SvbrdfDataset.mix`` (dataset.py:142-160) on the GPU (kernel K4): blend two materials -- normals projected
    to z = 1, blended and renormalised; diffuse, roughness, specular blended -- with weight ``alpha`` (one value,
    or one per batch item; drawn like the reference's when None).  [12,H,W] or [B,12,H,W] device tensors."""
    single = svbrdf_0.dim() == 3
    a = svbrdf_0.unsqueeze(0) if single else svbrdf_0
    b = svbrdf_1.unsqueeze(0) if single else svbrdf_1
    B = a.shape[0]
    if alpha is None:
        alpha = torch.cat([draw_mix_alpha() for _ in range(B)])
    alpha = torch.as_tensor(alpha, dtype=torch.float32).reshape(-1)
    if alpha.numel() == 1 and B > 1:
        alpha = alpha.expand(B)
    out = _native.mix_materials(a, b, alpha.to(a.device, non_blocking=True).contiguous())
    return out[0] if single else out
