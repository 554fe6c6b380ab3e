#!/usr/bin/env python3
"""(Re)writes svbrdf_estimation_amd/training/miopen_cache/MANIFEST.json after tools/install_miopen_cache.sh has unpacked a
cache built on an MI355X box: file hashes, the MIOpen build the cache belongs to, provenance.  tests/test_training_models.py
checks the tracked files against it; train.py prints its sha256 (config.miopen_cache).
    python tools/miopen_cache_manifest.py "how this cache was produced"
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "svbrdf_estimation_amd", "training", "miopen_cache")


def main():
    import torch
    files, build = {}, None
    for root, _, names in os.walk(D):
        for n in sorted(names):
            if n == "MANIFEST.json":
                continue
            p = os.path.join(root, n)
            with open(p, "rb") as f:
                files[os.path.relpath(p, D)] = {"bytes": os.path.getsize(p), "sha256": hashlib.sha256(f.read()).hexdigest()}
            m = re.match(r"gfx950\w*\.HIP\.(.+?)\.u?f?db\.txt$", n)
            if m:
                build = m.group(1)
    old = {}
    if os.path.exists(os.path.join(D, "MANIFEST.json")):
        with open(os.path.join(D, "MANIFEST.json")) as f:
            old = json.load(f)
    man = dict(old)
    man.update({"arch": "gfx950", "miopen_build": (build or "?") + " (from the db file names)", "torch": torch.__version__,
                "files": files})
    if len(sys.argv) > 1:
        man["produced_by"] = sys.argv[1]
    with open(os.path.join(D, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print(json.dumps(man["files"], indent=1))


if __name__ == "__main__":
    main()
