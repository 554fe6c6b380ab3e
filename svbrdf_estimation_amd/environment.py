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


def scenes_from_table(table):
    """[S,9] table -> list of Scene objects (what the reference's generators return)."""
    return [Scene(Camera(row[0:3]), Light(row[3:6], row[6:9].tolist())) for row in table]


def generate_random_scenes(count):
    return scenes_from_table(random_scene_table(count))


def generate_specular_scenes(count):
    return scenes_from_table(specular_scene_table(count))


def _triple(v, what):
    t = torch.as_tensor(v, dtype=torch.float32).detach().reshape(-1).cpu()
    if t.numel() != 3:
        raise ValueError("%s must have 3 components, got %d" % (what, t.numel()))
    return t


def scene_to_row(scene):
    """any object with .camera.pos, .light.pos, .light.color (lists, ndarrays or tensors) -> [9]"""
    return torch.cat((_triple(scene.camera.pos, "camera.pos"), _triple(scene.light.pos, "light.pos"),
                      _triple(scene.light.color, "light.color")))
