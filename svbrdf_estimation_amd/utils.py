"""SVBRDF tensor helpers with the reference's names and semantics.

Mirrors development/multiImage_pytorch/utils.py (the hot path's data contract):
12-channel layout  normals(0:3) | diffuse(3:6) | roughness(6:9) | specular(9:12) on
dim -3 (utils.py:36-58), the 9-channel network encoding (utils.py:73-90), range maps
(utils.py:92-98), gamma (utils.py:30-34) and the cosine-hemisphere direction sampler
(utils.py:100-111).  These are cheap layout/host helpers and run as stock torch ops on
whatever device the tensor lives on; the per-pixel shading is in csrc/.
"""
import math
import random

import numpy as np
import torch

_CHANNELS_12 = (3, 3, 3, 3)     # normals, diffuse, roughness, specular
_CHANNELS_9 = (2, 3, 1, 3)      # normals xy, diffuse, roughness (1 ch), specular


def enable_deterministic_random_engine(seed=313):
    """utils.py:7-13 (the cudnn switches are no-ops for this engine but kept for parity)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def gamma_decode(images, gamma=2.2):
    return torch.pow(images, gamma)


def gamma_encode(images, gamma=2.2):
    return torch.pow(images, 1.0 / gamma)


def pack_svbrdf(normals, diffuse, roughness, specular):
    """-> [...,12,H,W]; works for single maps and batches alike (channel dim is -3)."""
    return torch.cat((normals, diffuse, roughness, specular), dim=-3)


def unpack_svbrdf(svbrdf, is_encoded=False):
    """[...,12,H,W] -> (n, d, r, s); with is_encoded the 9-channel layout 2|3|1|3."""
    sizes = _CHANNELS_9 if is_encoded else _CHANNELS_12
    if svbrdf.shape[-3] != sum(sizes):
        raise ValueError("expected %d channels on dim -3, got %d" % (sum(sizes), svbrdf.shape[-3]))
    # contiguous copies, like the reference's split+cat, so callers may modify them freely
    return tuple(part.clone() for part in torch.split(svbrdf, sizes, dim=-3))


def decode_svbrdf(svbrdf):
    """9-channel network output in [-1,1] -> 12 channels (utils.py:73-90):
    normals = normalize(3*nx, 3*ny, 1); roughness repeated to 3 channels."""
    nxy, diffuse, roughness, specular = torch.split(svbrdf, _CHANNELS_9, dim=-3)
    nxy = nxy * 3.0
    nz = torch.ones_like(nxy.narrow(-3, 0, 1))
    normals = torch.cat((nxy, nz), dim=-3)
    norm = torch.sqrt(torch.sum(torch.pow(normals, 2.0), dim=-3, keepdim=True))
    normals = normals / norm
    reps = [1] * svbrdf.dim()
    reps[-3] = 3
    return pack_svbrdf(normals, diffuse, roughness.repeat(reps), specular)


def encode_as_unit_interval(tensor):
    """[-1,1] -> [0,1]"""
    return (tensor + 1) / 2


def decode_from_unit_interval(tensor):
    """[0,1] -> [-1,1]"""
    return tensor * 2 - 1


def generate_normalized_random_direction(count, min_eps=0.001, max_eps=0.05):
    """Cosine-weighted hemisphere directions [count,3] (utils.py:100-111).

    Draw order on torch's global CPU generator is part of the contract (SURVEY 8a row
    a16): r1 ~ U(min_eps, 1-max_eps) [count,1], then r2 ~ U(0,1) [count,1]."""
    r1 = torch.empty(count, 1).uniform_(0.0 + min_eps, 1.0 - max_eps)
    r2 = torch.empty(count, 1).uniform_(0.0, 1.0)
    radius = torch.sqrt(r1)
    phi = (2 * math.pi) * r2
    return torch.cat((radius * torch.cos(phi), radius * torch.sin(phi), torch.sqrt(1.0 - radius ** 2)), dim=-1)
