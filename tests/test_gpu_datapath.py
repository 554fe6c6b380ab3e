"""The real data path end to end on the GPU (BASELINE configs[2]'s per-GPU workload and configs[0]'s shape):

    tiled PNG files on disk -> DataLoader workers (TiledPngDataset: the reference's SvbrdfDataset, dataset.py:46-140)
    -> 8-bit or float transport -> device decode -> K1 input-photo synthesis (dataset.py:162-221) -> U-Net (MIOpen)
    -> fused MixedLoss (K3) -> backward -> Adam            (the reference's loop: main.py:47-63, 104-118)

Every piece is pinned on its own elsewhere (reader vs the reference bit for bit: tests/test_dataset_golden.py; transports:
test_gpu_parity.py; kernels: everywhere); this file runs the WHOLE of it through ``train.run`` in both transports."""
import os
import shutil
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write_samples(directory, count, tile, photos, seed):
    """`count` tiled-PNG samples: smooth low-frequency materials + fine noise (so that the PNG decode is realistic)"""
    from svbrdf_estimation_amd.training import data
    os.makedirs(directory, exist_ok=True)
    g = torch.Generator().manual_seed(seed)

    def smooth(c):
        low = torch.rand(1, c, max(2, tile // 16), max(2, tile // 16), generator=g)
        return torch.nn.functional.interpolate(low, size=(tile, tile), mode="bilinear", align_corners=False)[0]

    for i in range(count):
        n = smooth(3) * 0.6 - 0.3
        n[2] = 1.0
        n = n / n.norm(dim=0, keepdim=True)
        d, s = smooth(3), smooth(3) * 0.5
        r = smooth(1).expand(3, tile, tile) * 0.8 + 0.1
        svbrdf = torch.cat((n, d, r, s), 0) + 0.0
        svbrdf[3:] = (svbrdf[3:] + torch.randn(9, tile, tile, generator=g) * 0.01).clamp(0, 1)
        shots = [(d * 0.8 + torch.randn(3, tile, tile, generator=g) * 0.01).clamp(0, 1) for _ in range(photos)]
        data.write_tiled_png(os.path.join(directory, "sample_%03d.png" % i), shots, svbrdf)


@pytest.mark.timeout(1200)
def test_tiled_png_training_path_end_to_end_in_both_transports(tmp_path):
    """configs[2]'s per-GPU workload: 16 samples of 288x288 tiles with ONE stored photo (Deschaintre's format), random
    256x256 crops, batch 8, two DataLoader workers, mixed loss, fourteen steps -- once with the default 8-bit transport
    (cropped pixels cross to the device and are decoded there) and once with ``--float-transport`` (decoded in the
    workers, the reference's way).  Same seeds => the same crops, the same scenes, the same network initialisation:
    every step's loss must agree between the transports (the decoded maps are bit-identical; the stored photo goes
    through ``pow(2.2)`` on the host in one and through a lookup table built with that ``pow`` in the other, which on
    some hosts differ in the last bit: 1e-5 relative is far below what any real divergence would show).  The loss must
    be finite throughout, and training on the fixed 16 materials must reduce it."""
    import train
    d = str(tmp_path / "pngs")
    _write_samples(d, 16, 288, 1, seed=5)
    common = ["--data", d, "--image-count", "1", "--batch", "8", "--workers", "2", "--random-crop", "--steps", "14",
              "--warmup", "0", "--conv-mode", "hybrid", "--lr", "2e-4"]
    runs = {}
    for name, extra in (("uint8", []), ("float", ["--float-transport"])):
        res = train.run(train.parse_args(common + extra))
        assert res["config"]["data"] == "tiled-png" and res["config"]["workers"] == 2 and res["config"]["per_gpu_batch"] == 8
        per_step = np.array(res["loss_per_step"])
        assert len(per_step) == 14 and np.isfinite(per_step).all(), per_step
        runs[name] = per_step
        print("[datapath] %s transport: %.1f patches/s, loss %s" % (name, res["value"], np.round(per_step, 4).tolist()))
    a, b = runs["uint8"], runs["float"]
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]), (a[0], b[0])               # first step: identical inputs and weights
    assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max(), (a, b)            # later steps: fp32 noise through 13 Adam updates
    for per_step in runs.values():
        assert per_step[-4:].mean() < per_step[:4].mean(), per_step        # it learns the 16 materials


@pytest.mark.timeout(1200)
def test_toy_sample_batch_one_goes_through_the_engine(tmp_path):
    """configs[0]'s shape through the HIP path: the reference's bundled toy sample format (10 stored photos + 4 maps per
    PNG, ``--image-count 10``; tests/golden/g12_tiled_toy_crop.png holds the top-left 64x64 of each tile of that
    sample), batch 1, single view, two steps.  The eight-level U-Net needs 256x256, so the 64x64 tiles go through the
    reader's 'resize' scale mode (dataset.py:58-73: centre crop, bilinear resize; float transport).  The reference runs
    this configuration on the CPU (train.sh:8); here the stored photo is uploaded and the maps meet the fused loss on
    the GPU."""
    import train
    d = str(tmp_path / "toy")
    os.makedirs(d)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "g12_tiled_toy_crop.png"), d)
    res = train.run(train.parse_args(["--data", d, "--image-count", "10", "--scale-mode", "resize", "--size", "256", "--batch", "1",
                                      "--workers", "0", "--steps", "2", "--warmup", "0", "--conv-mode", "hybrid"]))
    assert res["config"]["data"] == "tiled-png" and res["config"]["per_gpu_batch"] == 1 and res["config"]["size"] == 256
    per_step = np.array(res["loss_per_step"])
    assert len(per_step) == 2 and np.isfinite(per_step).all() and (per_step > 0).all(), per_step
