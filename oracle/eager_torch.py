"""Eager-PyTorch restatement of the reference's execution model -- TEST INFRASTRUCTURE.

What bench.py times as ``cpu_baseline`` (kind "port"): the reference computes the hot path
as a Python double loop (batch item x scene) of small eager tensor ops -- one op per
arithmetic step of renderers.py:67-104, then log/L1 (losses.py:29-52) and autograd for
the backward.  The reference's files cannot travel to the GPU box, so this module
restates that algorithm with stock torch ops in the same arithmetic order; on the same
torch build it reproduces the reference bit for bit (tests/test_oracle_golden.py checks
it against the fixtures), which also makes it a second, independent oracle.

Never imported by the product package.
"""
import math

import torch


def _dot(a, b):
    # renderers.py:8-9
    return (a * b).sum(dim=-3, keepdim=True)


def _unit(v):
    # renderers.py:11-12
    return v / torch.sqrt(_dot(v, v))


def _clamp_dot(a, b):
    return torch.clamp(_dot(a, b), min=0.001)


def _smith_g1(alpha_sq, cos_sq):
    # renderers.py:34-38 with xi == 1 (its argument is a ratio of values clamped >= 1e-3)
    return 2.0 / (1 + torch.sqrt(1 + alpha_sq * (1.0 - cos_sq) / cos_sq))


def patch_coords(H, W, device=None):
    """renderers.py:73-76: x along columns, y = -x along rows, z = 0"""
    xs = torch.linspace(-1, 1, W, device=device)
    gx = xs.unsqueeze(0).expand(H, W).unsqueeze(0)
    gy = -1 * gx.transpose(1, 2)
    return torch.cat((gx, gy, torch.zeros_like(gx)), dim=0)


def render_scene(svbrdf, scene_row):
    """one render of renderers.py:67-104.  svbrdf [12,H,W] or [B,12,H,W]; scene_row [9]."""
    pos = patch_coords(svbrdf.shape[-2], svbrdf.shape[-1], svbrdf.device)
    cam = scene_row[0:3].reshape(3, 1, 1)
    lgt = scene_row[3:6].reshape(3, 1, 1)
    col = scene_row[6:9].reshape(1, 3, 1, 1)
    n, kd, rough, ks0 = torch.split(svbrdf, (3, 3, 3, 3), dim=-3)
    rough = torch.clamp(rough, min=0.001)
    to_cam = cam - pos
    wo = _unit(to_cam)
    to_light = lgt - pos
    wi = _unit(to_light)
    half = _unit((wi + wo) / 2.0)
    n_h, v_h = _clamp_dot(n, half), _clamp_dot(wo, half)
    v_n, l_n = _clamp_dot(wo, n), _clamp_dot(wi, n)
    fresnel = ks0 + (1.0 - ks0) * (1.0 - v_h) ** 5
    a2 = (rough ** 2) ** 2
    geom = _smith_g1(a2, v_n ** 2) * _smith_g1(a2, l_n ** 2)
    nh2 = n_h ** 2
    den = torch.clamp(nh2 * (a2 + (1 - nh2) / nh2), min=0.001)
    ggx = a2 / (math.pi * den ** 2)
    specular = fresnel * geom * ggx / (4.0 * v_n * l_n)
    diffuse = (1.0 - fresnel) * kd / math.pi
    cos_l = torch.clamp(_dot(wi, n), min=0.0)
    falloff = 1.0 / torch.sqrt(_dot(to_light, to_light)) ** 2
    return ((diffuse + specular) * (col * falloff)) * cos_l


def rendering_loss(input, target, scene_table, eps=0.1):
    """losses.py:29-52 with the scenes given as a host table [B,S,9]."""
    ins, tgs = [], []
    for b in range(input.shape[0]):
        ri = [render_scene(input[b], scene_table[b, s]) for s in range(scene_table.shape[1])]
        rt = [render_scene(target[b], scene_table[b, s]) for s in range(scene_table.shape[1])]
        ins.append(torch.cat(ri, dim=0))
        tgs.append(torch.cat(rt, dim=0))
    a = torch.log(torch.stack(ins, dim=0) + eps)
    t = torch.log(torch.stack(tgs, dim=0) + eps)
    return torch.nn.functional.l1_loss(a, t)
