"""Scene value types and the light/view sampler of the rendering loss.

Mirrors development/multiImage_pytorch/environment.py: ``Camera``/``Light``/``Scene``
attribute bags (:4-16) and ``generate_random_scenes`` / ``generate_specular_scenes``
(:18-55).  Host-side code on torch's global CPU generator; the draw ORDER is part of
the contract so that a seeded training run samples the same scenes as the reference:

  per RenderingLoss batch item (losses.py:35):
    random(n):    view dirs  U[n,1] U[n,1]  ->  light dirs  U[n,1] U[n,1]
    specular(m):  view dirs  U[m,1] U[m,1]  ->  N[m,1] (view distance)
                  -> N[m,1] (light distance) -> U[m,2] (shift)

The engine consumes scenes as one fp32 table [S,9] = camera xyz | light xyz | light rgb
(`*_scene_table`), which is what the kernels read through scalar loads.
"""
import math

import torch

from . import utils

RANDOM_LIGHT_POWER = 20.0      # environment.py:27
SPECULAR_LIGHT_POWER = 50.0    # environment.py:52


class Camera:
    def __init__(self, pos):
        self.pos = pos


class Light:
    def __init__(self, pos, color):
        self.pos = pos
        self.color = color


class Scene:
    def __init__(self, camera, light):
        self.camera = camera
        self.light = light


def random_scene_table(count):
    """[count,9]: independent cosine-hemisphere view and light directions at distance 1."""
    view = utils.generate_normalized_random_direction(count, 0.001, 0.1)
    light = utils.generate_normalized_random_direction(count, 0.001, 0.1)
    color = torch.full((count, 3), RANDOM_LIGHT_POWER)
    return torch.cat((view, light, color), dim=-1)


def specular_scene_table(count):
    """[count,9]: light mirrored about the patch normal, log-normal distances, common xy shift."""
    view = utils.generate_normalized_random_direction(count, 0.001, 0.1)
    light = view * torch.tensor([[-1.0, -1.0, 1.0]])
    view_distance = torch.exp(torch.empty(count, 1).normal_(mean=0.5, std=0.75))
    light_distance = torch.exp(torch.empty(count, 1).normal_(mean=0.5, std=0.75))
    shift = torch.cat((torch.empty(count, 2).uniform_(-1.0, 1.0), torch.zeros(count, 1) + 0.0001), dim=-1)
    view = view * view_distance + shift
    light = light * light_distance + shift
    color = torch.full((count, 3), SPECULAR_LIGHT_POWER)
    return torch.cat((view, light, color), dim=-1)


def scene_table(random_count, specular_count):
    """scenes of ONE batch item, [random_count+specular_count, 9], reference draw order."""
    parts = []
    if random_count > 0:
        parts.append(random_scene_table(random_count))
    if specular_count > 0:
        parts.append(specular_scene_table(specular_count))
    if not parts:
        return torch.zeros(0, 9)
    return torch.cat(parts, dim=0)


class BatchSceneSampler:
    """[B,S,9] scene tables for a whole batch with the reference's RNG draw order.

    The reference draws item after item (losses.py:32-35).  The global CPU generator is
    consumed here in exactly the same element order, but with fewer, larger calls into
    cached views of batch-wide buffers: per item one ``uniform_(0,1)`` covers the six
    consecutive uniform draws of the direction sampler (a ``uniform_(lo,hi)`` is
    ``fma(u, hi-lo, lo)`` of the same raw 24-bit ``u`` -- reproduced with ``addcmul``), then
    the two ``normal_`` calls and the shift ``uniform_(-1,1)`` as in the reference.  The
    trigonometry / exponentials run once over the whole batch instead of once per item.
    tests/test_host_logic.py checks tables AND the generator state bit for bit against
    the per-item functions above.
    """

    def __init__(self, batch_size, random_count, specular_count):
        B, R, M = int(batch_size), int(random_count), int(specular_count)
        self.shape = (B, R, M)
        self._raw = torch.empty(B, 4 * R + 2 * M)      # per item: r1v r2v r1l r2l (random) | r1 r2 (specular view)
        idx = torch.arange(4 * R + 2 * M)
        self._i1 = torch.cat((idx[0:R], idx[2 * R:3 * R], idx[4 * R:4 * R + M]))
        self._i2 = torch.cat((idx[R:2 * R], idx[3 * R:4 * R], idx[4 * R + M:]))
        lo, hi = torch.tensor(0.0 + 0.001), torch.tensor(1.0 - 0.1)    # environment.py:20-21 -> utils.py:101
        self._lo, self._width = lo, hi - lo
        self._nrm = torch.empty(B, 2, M)               # log-distances: view, light
        self._shift = torch.empty(B, M, 3)
        self._shift[:, :, 2] = torch.zeros(B, M) + 0.0001
        self._mirror = torch.tensor([-1.0, -1.0, 1.0])
        self._col_r = torch.full((B, R, 3), RANDOM_LIGHT_POWER)
        self._col_s = torch.full((B, M, 3), SPECULAR_LIGHT_POWER)
        self._draws = []                               # bound in-place RNG calls, reference order
        self._shift_buf = torch.empty(B, M, 2)
        for b in range(B):
            self._draws.append((self._raw[b].uniform_, 0.0, 1.0))
            if M > 0:
                if 2 * M < 16:
                    # below 16 elements torch's CPU normal_ is the sequential scalar Box-Muller path
                    # (its cached second sample lives in the generator), so ONE call over the 2M
                    # contiguous slots draws exactly what the reference's two M-element calls draw
                    self._draws.append((self._nrm[b].normal_, 0.5, 0.75))
                else:
                    self._draws.append((self._nrm[b, 0].normal_, 0.5, 0.75))
                    self._draws.append((self._nrm[b, 1].normal_, 0.5, 0.75))
                self._draws.append((self._shift_buf[b].uniform_, -1.0, 1.0))

    def sample(self):
        B, R, M = self.shape
        for fn, a, b in self._draws:
            fn(a, b)
        r1 = torch.addcmul(self._lo, self._raw.index_select(1, self._i1), self._width)
        r2 = self._raw.index_select(1, self._i2)
        radius = torch.sqrt(r1)
        phi = (2 * math.pi) * r2
        dirs = torch.stack((radius * torch.cos(phi), radius * torch.sin(phi), torch.sqrt(1.0 - radius ** 2)), dim=-1)
        parts = []
        if R > 0:
            parts.append(torch.cat((dirs[:, :R], dirs[:, R:2 * R], self._col_r), dim=-1))
        if M > 0:
            view = dirs[:, 2 * R:]
            dist = torch.exp(self._nrm).unsqueeze(-1)                      # [B,2,M,1]
            self._shift[:, :, 0:2] = self._shift_buf
            parts.append(torch.cat((view * dist[:, 0] + self._shift, (view * self._mirror) * dist[:, 1] + self._shift,
                                    self._col_s), dim=-1))
        if not parts:
            return torch.zeros(B, 0, 9)
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


def scenes_from_table(table):
    """[S,9] table -> list of Scene objects (what the reference's generators return)."""
    return [Scene(Camera(row[0:3]), Light(row[3:6], row[6:9].tolist())) for row in table]


def generate_random_scenes(count):
    return scenes_from_table(random_scene_table(count))


def generate_specular_scenes(count):
    return scenes_from_table(specular_scene_table(count))


def _triple(v, what):
    """3-sequence, ndarray or tensor -> three python floats (ONE host read for a tensor, whatever device it is on)"""
    if isinstance(v, torch.Tensor):
        vals = v.detach().reshape(-1).tolist()
    elif isinstance(v, (list, tuple)):
        vals = [float(x) for x in v]
    else:
        import numpy as np
        vals = np.asarray(v, dtype=np.float64).reshape(-1).tolist()
    if len(vals) != 3:
        raise ValueError("%s must have 3 components, got %d" % (what, len(vals)))
    return vals


def scene_floats(scene):
    """any object with .camera.pos, .light.pos, .light.color -> nine python floats (camera xyz | light xyz | rgb)"""
    return _triple(scene.camera.pos, "camera.pos") + _triple(scene.light.pos, "light.pos") + _triple(scene.light.color, "light.color")


def scene_to_row(scene):
    """any object with .camera.pos, .light.pos, .light.color (lists, ndarrays or tensors) -> host fp32 [9]
    (float32 rounding of the values as ``torch.Tensor(...)`` does it, renderers.py:79,91,98)"""
    return torch.tensor(scene_floats(scene), dtype=torch.float32)
