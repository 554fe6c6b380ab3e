#!/usr/bin/env python3
"""Samples/s of the tiled-PNG reader (svbrdf_estimation_amd/training/data.py, the reference's SvbrdfDataset format) per
DataLoader worker count -- what sizes train.py's --workers against the step rate.  CPU only, no GPU needed.
    python3 tools/loader_rate.py [--tile 288] [--photos 1] [--samples 64] [--workers 0 1 2 4 8]
Writes N synthetic tiled PNGs (photos + 4 map tiles of tile x tile, Deschaintre layout) to a temp dir and times one pass
per worker count (batch 8, random 256-crop as main.py:49 uses for training)."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=288)
    ap.add_argument("--photos", type=int, default=1)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--workers", type=int, nargs="+", default=[0, 1, 2, 4, 8])
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    from PIL import Image
    from svbrdf_estimation_amd.training import data
    d = tempfile.mkdtemp()
    rng = np.random.RandomState(0)
    # smooth-ish content (PNG decode time depends on compressibility): low-res noise upsampled + fine noise
    for i in range(a.samples):
        tiles = a.photos + 4
        low = rng.randint(0, 256, size=(a.tile // 8, tiles * a.tile // 8, 3)).astype(np.uint8)
        img = np.repeat(np.repeat(low, 8, 0), 8, 1).astype(np.int16) + rng.randint(-6, 7, size=(a.tile, tiles * a.tile, 3))
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(d, "%04d.png" % i))
    size_mb = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e6
    ds = data.TiledPngDataset(d, image_size=256, image_count=a.photos, used_image_count=1, random_crop=True)
    out = {"tile": a.tile, "photos_stored": a.photos, "samples": a.samples, "png_MB_total": size_mb, "cpus": os.cpu_count(),
           "batch": a.batch, "rates": {}}
    for w in a.workers:
        loader = torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=True, num_workers=w, drop_last=True,
                                             persistent_workers=w > 0)
        for _ in loader:            # first pass: worker start-up, file cache
            pass
        t0, n = time.perf_counter(), 0
        for _ in range(2):
            for b in loader:
                n += b["svbrdf"].shape[0]
        out["rates"][str(w)] = n / (time.perf_counter() - t0)
        del loader
    print(json.dumps(out))


if __name__ == "__main__":
    main()
