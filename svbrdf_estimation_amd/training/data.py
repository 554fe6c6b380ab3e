"""Samples for training: the Deschaintre tiled-PNG format of the reference and a synthetic source.

Tiled PNG (development/multiImage_pytorch/dataset.py:105-140): one image per material,
``image_count`` input photos followed by the four maps normals | diffuse | roughness | specular,
all the same size, laid side by side along the width; normals are stored in [0,1] and mapped to
[-1,1] on load (:128); the LAST ``used`` photos are the ones read (:136-138).  Photos that are not
stored are synthesised -- here on the GPU, for the whole batch at once, with
``svbrdf_estimation_amd.synthesis.render_inputs`` (row f3) instead of inside the CPU dataloader.
"""
import os

import numpy as np
import torch

from .. import synthesis, utils


def read_tiled_png(path, image_count):
    """-> (photos [n,3,H,W] in [0,1] as stored, svbrdf [12,H,W]) from one tiled sample."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0      # [H, (n+4)*H, 3]
    full = torch.from_numpy(img).permute(2, 0, 1)
    tiles = image_count + 4
    if full.shape[-1] % tiles != 0:
        raise ValueError("%s: width %d is not %d equal tiles" % (path, full.shape[-1], tiles))
    parts = torch.stack(full.chunk(tiles, dim=-1), dim=0)                             # [tiles,3,H,W]
    normals = utils.decode_from_unit_interval(parts[image_count])
    svbrdf = utils.pack_svbrdf(normals, parts[image_count + 1], parts[image_count + 2], parts[image_count + 3])
    return parts[:image_count], svbrdf.contiguous()


def write_tiled_png(path, photos, svbrdf):
    """inverse of read_tiled_png (8-bit), used by the tests and to export synthetic samples"""
    from PIL import Image
    n, d, r, s = torch.split(svbrdf, (3, 3, 3, 3), dim=-3)
    tiles = [p for p in photos] + [utils.encode_as_unit_interval(n), d, r, s]
    row = torch.cat(tiles, dim=-1).clamp(0, 1).permute(1, 2, 0).numpy()
    Image.fromarray(np.uint8(np.round(row * 255.0))).save(path)


class TiledPngDataset(torch.utils.data.Dataset):
    """{'inputs': stored photos (gamma-decoded unless linear) [n,3,S,S], 'svbrdf': [12,S,S]}; missing photos
    are added later on the GPU by ``complete_inputs``."""

    def __init__(self, directory, image_size=256, image_count=10, used_image_count=1, is_linear=False):
        self.paths = sorted(os.path.join(directory, f) for f in os.listdir(directory)
                            if os.path.isfile(os.path.join(directory, f)))
        self.image_size, self.image_count, self.used, self.is_linear = image_size, image_count, used_image_count, is_linear

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        photos, svbrdf = read_tiled_png(self.paths[idx], self.image_count)
        keep = min(self.image_count, self.used)
        photos = photos[self.image_count - keep:]                                     # the last ones, dataset.py:137
        S = self.image_size
        photos, svbrdf = photos[..., :S, :S], svbrdf[..., :S, :S]                      # scale_mode 'crop', anchor 0
        if not self.is_linear:
            photos = utils.gamma_decode(photos)
        return {"inputs": photos.contiguous(), "svbrdf": svbrdf.contiguous()}


class SyntheticSvbrdfDataset(torch.utils.data.Dataset):
    """random smooth-ish SVBRDFs (BASELINE.md section 3 statistics); no stored photos"""

    def __init__(self, length, image_size=256, seed=0):
        self.length, self.size, self.seed = length, image_size, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        H = self.size
        n = torch.randn(3, H, H, generator=g) * 0.3
        n[2] = 1.0 + n[2].abs()
        n = n / n.norm(dim=0, keepdim=True)
        d = torch.rand(3, H, H, generator=g)
        r = torch.rand(1, H, H, generator=g).expand(3, H, H)
        s = torch.rand(3, H, H, generator=g)
        return {"inputs": torch.zeros(0, 3, H, H), "svbrdf": torch.cat((n, d, r, s), dim=0).contiguous()}


def complete_inputs(batch_inputs, batch_svbrdf, used_image_count, use_augmentation=True, noise="device"):
    """device tensors: [B,n,3,H,W] stored photos (n may be 0) + [B,12,H,W] maps -> [B,used,3,H,W], the missing
    ones rendered by one launch of K1 for the whole batch (dataset.py:94-98 did this per sample on the CPU)."""
    missing = used_image_count - batch_inputs.shape[1]
    if missing <= 0:
        return batch_inputs[:, :used_image_count]
    rendered = synthesis.render_inputs(batch_svbrdf, missing, use_augmentation=use_augmentation, noise=noise)
    return torch.cat((batch_inputs, rendered), dim=1)
