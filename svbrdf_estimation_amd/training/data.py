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


def _crop_square(t, anchor, size):
    """utils.crop_square of the reference (utils.py:15-29) for one anchor: rows anchor[0].., columns anchor[1].."""
    return t[..., anchor[0]:anchor[0] + size, anchor[1]:anchor[1] + size]


class TiledPngDataset(torch.utils.data.Dataset):
    """The reference's ``SvbrdfDataset`` (dataset.py:11-140), same constructor meaning:

    ``scale_mode``  'crop' -- a ``image_size`` window at the top-left corner, or at a random anchor with
                    ``random_crop`` (two ``np.random.randint`` draws: row, then column, dataset.py:82-83);
                    'resize' -- centre square crop, then bilinear resize to ``image_size`` (dataset.py:58-73).
    ``image_count`` photos stored per sample, ``used_image_count`` photos wanted: the LAST
                    ``min(image_count, used)`` stored ones are read (dataset.py:136-138), gamma-decoded unless
                    ``is_linear``; missing ones are rendered later on the GPU (``complete_inputs``).
    ``mix_materials`` only for ``image_count == 0`` (dataset.py:29-32): a partner sample is picked with python's
                    ``random.randrange`` and a blend weight with ``torch`` U(0.1, 0.9) (dataset.py:52-56, :144) --
                    same generators, same order -- and returned as ``svbrdf_other`` / ``mix_alpha``; the blend
                    itself runs on the GPU for the whole batch (``apply_mixing`` -> kernel K4).  The reference
                    blends at full resolution BEFORE it scales (dataset.py:52-73).  A crop commutes with the
                    per-pixel blend exactly, a bilinear resize does not (the blended normals are renormalised), so
                    with 'resize' the item carries both materials centre-cropped but UNRESIZED plus ``resize_to``,
                    and ``apply_mixing`` blends at that resolution and resizes afterwards, in the reference's order.
    ``no_svbrdf``   photos only: a flat dummy SVBRDF (normals (0,0,1), everything else 0), dataset.py:116-124.

    ``uint8_transport`` ('crop' mode): items hold the cropped 8-bit pixels ('inputs_u8' [n,3,S,S], 'svbrdf_u8' [12,S,S],
                    'svbrdf_other_u8') instead of decoded floats; ``decode_uint8_batch`` turns a collated batch into
                    exactly the tensors of the float path, on whatever device it is given.  Same RNG draws, same order.

    Returns {'inputs': [n,3,S,S], 'svbrdf': [12,S,S]} (+ 'svbrdf_other', 'mix_alpha' when mixing)."""

    def __init__(self, directory, image_size=256, image_count=10, used_image_count=1, is_linear=False, scale_mode="crop",
                 random_crop=False, mix_materials=False, no_svbrdf=False, uint8_transport=False):
        self.paths = sorted(os.path.join(directory, f) for f in os.listdir(directory)
                            if os.path.isfile(os.path.join(directory, f)))
        if scale_mode not in ("crop", "resize"):
            raise ValueError("Unknown scale mode {}".format(scale_mode))
        self.image_size, self.image_count, self.used, self.is_linear = image_size, image_count, used_image_count, is_linear
        self.scale_mode, self.random_crop, self.no_svbrdf = scale_mode, random_crop, no_svbrdf
        # uint8_transport: hand the cropped 8-bit pixels to the training process and decode them THERE, on the device
        # (decode_uint8_batch: three 256-entry lookup tables built with the reference's CPU arithmetic, so the decoded
        # tensors are bit for bit what this reader returns otherwise) -- a quarter of the bytes through the worker ->
        # pinned-memory -> device pipeline and no float conversion in the workers.  'crop' mode only (a resize needs floats).
        self.uint8_transport = bool(uint8_transport) and scale_mode == "crop" and not no_svbrdf
        self.mix_materials = mix_materials
        if self.mix_materials and self.image_count > 0:
            self.mix_materials = False
            print("Warning: Material mixing is only supported for datasets without input images.")

    def __len__(self):
        return len(self.paths)

    def read_sample(self, path):
        """-> (the last min(image_count, used) stored photos as stored, svbrdf [12,H,W]); dataset.py:105-140"""
        if self.no_svbrdf:
            from PIL import Image
            img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
            parts = torch.stack(torch.from_numpy(img).permute(2, 0, 1).chunk(self.image_count, dim=-1), dim=0)
            h, w = parts.shape[-2:]
            normals = torch.cat((torch.zeros(2, h, w), torch.ones(1, h, w)), dim=0)
            svbrdf = torch.cat((normals, torch.zeros(9, h, w)), dim=0)
            photos = parts
        else:
            photos, svbrdf = read_tiled_png(path, self.image_count)
        keep = min(self.image_count, self.used)
        return photos[self.image_count - keep:self.image_count], svbrdf

    def _scale(self, photos, svbrdfs, resize_maps=True):
        """the reference's crop / resize of the photos and of every SVBRDF in `svbrdfs` with ONE anchor;
        ``resize_maps=False`` stops after the centre crop of the maps (material mixing comes before the resize)"""
        height, width = svbrdfs[0].shape[-2:]
        S = self.image_size
        if self.scale_mode == "resize":
            landscape = width > height
            anchor = (0, (width - height) // 2) if landscape else ((height - width) // 2, 0)
            size = height if landscape else width
            photos = _crop_square(photos, anchor, size)
            svbrdfs = [_crop_square(m, anchor, size) for m in svbrdfs]
            interp = torch.nn.functional.interpolate
            if photos.shape[0] > 0:
                photos = interp(photos, size=(S, S), mode="bilinear")
            else:
                photos = photos.new_zeros((0, 3, S, S))
            if resize_maps:
                svbrdfs = [interp(m.unsqueeze(0), size=(S, S), mode="bilinear").squeeze(0) for m in svbrdfs]
        else:
            anchor = (0, 0)
            if self.random_crop:
                anchor = (np.random.randint(0, height - S + 1), np.random.randint(0, width - S + 1))
            photos = _crop_square(photos, anchor, S)
            svbrdfs = [_crop_square(m, anchor, S) for m in svbrdfs]
        return photos, svbrdfs

    def _read_tiles_u8(self, path):
        """-> uint8 [tiles,3,H,W]: the sample's tiles as stored (photos, then normals | diffuse | roughness | specular)"""
        from PIL import Image
        img = torch.from_numpy(np.array(Image.open(path).convert("RGB"))).permute(2, 0, 1)       # [3,H,tiles*W] uint8
        tiles = self.image_count + 4
        if img.shape[-1] % tiles != 0:
            raise ValueError("%s: width %d is not %d equal tiles" % (path, img.shape[-1], tiles))
        return torch.stack(img.chunk(tiles, dim=-1), dim=0)

    def _getitem_u8(self, idx):
        """the 'crop' path on 8-bit pixels: same draws in the same order as __getitem__ (partner, weight, anchor)"""
        tiles = self._read_tiles_u8(self.paths[idx])
        other, alpha = None, None
        if self.mix_materials:
            import random
            other = self._read_tiles_u8(self.paths[random.randrange(0, len(self))])     # dataset.py:54
            alpha = synthesis.draw_mix_alpha()                                          # dataset.py:144
        height, width = tiles.shape[-2:]
        S = self.image_size
        anchor = (0, 0)
        if self.random_crop:
            anchor = (np.random.randint(0, height - S + 1), np.random.randint(0, width - S + 1))
        keep = min(self.image_count, self.used)
        n = self.image_count
        crop = lambda t: _crop_square(t, anchor, S).contiguous()
        item = {"inputs_u8": crop(tiles[n - keep:n]), "svbrdf_u8": crop(tiles[n:n + 4]).reshape(12, S, S)}
        if other is not None:
            item["svbrdf_other_u8"], item["mix_alpha"] = crop(other[n:n + 4]).reshape(12, S, S), alpha
        return item

    def __getitem__(self, idx):
        if self.uint8_transport:
            return self._getitem_u8(idx)
        photos, svbrdf = self.read_sample(self.paths[idx])
        maps, alpha = [svbrdf], None
        if self.mix_materials:
            import random
            other = random.randrange(0, len(self))                                    # dataset.py:54
            maps.append(self.read_sample(self.paths[other])[1])
            alpha = synthesis.draw_mix_alpha()                                        # dataset.py:144
        blend_first = self.mix_materials and self.scale_mode == "resize"       # dataset.py:52-73: mix, THEN resize
        photos, maps = self._scale(photos, maps, resize_maps=not blend_first)
        if not self.is_linear:
            photos = utils.gamma_decode(photos)
        item = {"inputs": photos.contiguous(), "svbrdf": maps[0].contiguous()}
        if self.mix_materials:
            item["svbrdf_other"], item["mix_alpha"] = maps[1].contiguous(), alpha
            if blend_first:
                item["resize_to"] = self.image_size
        return item


_LUTS = {}


def _decode_luts(device):
    """three 256-entry tables, computed ONCE on the CPU with the float path's own operations -- value / 255 as
    ``read_tiled_png`` forms it, ``* 2 - 1`` for normals (utils.decode_from_unit_interval), ``pow(2.2)`` for stored photos
    (utils.gamma_decode) -- so that a gather on any device reproduces the float path: exactly for the maps (IEEE division
    and multiply-add of 256 values), and for the photos up to the last bit of torch's CPU ``pow`` (not correctly rounded:
    its vectorised and scalar code paths can differ by 1 ULP on some hosts; identical in the build container)"""
    key = str(device)
    if key not in _LUTS:
        unit = torch.from_numpy(np.arange(256, dtype=np.uint8).astype(np.float32) / 255.0)
        _LUTS[key] = tuple(t.to(device) for t in (unit, utils.decode_from_unit_interval(unit), utils.gamma_decode(unit)))
    return _LUTS[key]


def decode_uint8_batch(batch, device, is_linear=False):
    """collated batch of a ``uint8_transport`` dataset -> the batch the float path would have produced, on `device`:
    {'inputs' [B,n,3,S,S], 'svbrdf' [B,12,S,S]} (+ 'svbrdf_other', 'mix_alpha').  A float batch passes through."""
    if "svbrdf_u8" not in batch:
        return batch
    unit, normals, gamma = _decode_luts(torch.device(device))

    def maps(u8):
        idx = u8.to(device, non_blocking=True).long()
        return torch.cat((normals[idx[:, 0:3]], unit[idx[:, 3:12]]), dim=1)

    photos = batch["inputs_u8"].to(device, non_blocking=True).long()
    out = {"inputs": (unit if is_linear else gamma)[photos], "svbrdf": maps(batch["svbrdf_u8"])}
    if "svbrdf_other_u8" in batch:
        out["svbrdf_other"], out["mix_alpha"] = maps(batch["svbrdf_other_u8"]), batch["mix_alpha"]
    return out


def apply_mixing(batch_svbrdf, batch):
    """device side of the material-mixing augmentation: [B,12,H,W] maps + the collated batch of a mixing dataset
    ('svbrdf_other' [B,12,H,W], 'mix_alpha' [B,1]) -> mixed maps (kernel K4).  A batch without those keys passes."""
    if "svbrdf_other" not in batch:
        return batch_svbrdf
    other = batch["svbrdf_other"].to(batch_svbrdf.device, non_blocking=True)
    mixed = synthesis.mix_materials(batch_svbrdf, other, batch["mix_alpha"].reshape(-1))
    if "resize_to" in batch:            # scale_mode 'resize': the pair came centre-cropped at full resolution
        S = int(torch.as_tensor(batch["resize_to"]).reshape(-1)[0])
        mixed = torch.nn.functional.interpolate(mixed, size=(S, S), mode="bilinear")
    return mixed


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


def synthetic_svbrdf_batch(batch, image_size, device, generator):
    """[B,12,H,W] random SVBRDF maps with SyntheticSvbrdfDataset's statistics, drawn on `device` with `generator`
    (train.py --data synthetic: no dataloader in the way of the step rate)"""
    H = image_size
    n = torch.randn(batch, 3, H, H, device=device, generator=generator) * 0.3
    n[:, 2] = 1.0 + n[:, 2].abs()
    n = n / n.norm(dim=1, keepdim=True)
    d = torch.rand(batch, 3, H, H, device=device, generator=generator)
    r = torch.rand(batch, 1, H, H, device=device, generator=generator).expand(batch, 3, H, H)
    s = torch.rand(batch, 3, H, H, device=device, generator=generator)
    return torch.cat((n, d, r, s), dim=1)


def complete_inputs(batch_inputs, batch_svbrdf, used_image_count, use_augmentation=True, noise="device"):
    """device tensors: [B,n,3,H,W] stored photos (n may be 0) + [B,12,H,W] maps -> [B,used,3,H,W], the missing
    ones rendered by one launch of K1 for the whole batch (dataset.py:94-98 did this per sample on the CPU)."""
    missing = used_image_count - batch_inputs.shape[1]
    if missing <= 0:
        return batch_inputs[:, :used_image_count]
    rendered = synthesis.render_inputs(batch_svbrdf, missing, use_augmentation=use_augmentation, noise=noise)
    return torch.cat((batch_inputs, rendered), dim=1)
