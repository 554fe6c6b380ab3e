#!/usr/bin/env python3
"""Writes N synthetic samples in the reference's tiled-PNG format (development/multiImage_pytorch/dataset.py:105-140:
`photos` input tiles followed by normals | diffuse | roughness | specular, side by side; Deschaintre's data: 288x288
tiles) for end-to-end runs of train.py with the real reader.  Smooth low-frequency content + fine noise, so that the PNG
decode cost is realistic.   python3 tools/make_png_dataset.py DIR [--samples 64] [--tile 288] [--photos 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--tile", type=int, default=288)
    ap.add_argument("--photos", type=int, default=1)
    a = ap.parse_args()
    from svbrdf_estimation_amd.training import data
    os.makedirs(a.dir, exist_ok=True)
    g = torch.Generator().manual_seed(0)
    T = a.tile

    def smooth(c):
        low = torch.rand(1, c, T // 16, T // 16, generator=g)
        return torch.nn.functional.interpolate(low, size=(T, T), mode="bilinear", align_corners=False)[0]

    for i in range(a.samples):
        n = smooth(3) * 0.6 - 0.3
        n[2] = 1.0
        n = n / n.norm(dim=0, keepdim=True)
        d, s = smooth(3), smooth(3) * 0.5
        r = smooth(1).expand(3, T, T) * 0.8 + 0.1
        svbrdf = torch.cat((n, d, r, s), 0) + 0.0
        svbrdf[3:] = (svbrdf[3:] + torch.randn(9, T, T, generator=g) * 0.01).clamp(0, 1)
        photos = [(d * 0.8 + torch.randn(3, T, T, generator=g) * 0.01).clamp(0, 1) for _ in range(a.photos)]
        data.write_tiled_png(os.path.join(a.dir, "%05d.png" % i), photos, svbrdf)
    print("wrote %d samples of %d tiles %dx%d to %s" % (a.samples, a.photos + 4, T, T, a.dir))


if __name__ == "__main__":
    main()
