"""what sits either side of the hot path in a training run -- the
Deschaintre U-Net (stock PyTorch-ROCm modules, no custom kernels), the tiled-PNG sample reader,
and a one-process-per-GPU DDP harness (RCCL all-reduce of the U-Net gradients; the rendering
loss itself shards by batch with no collective).  See train.py at the repository root."""
from . import data, models  # noqa: F401


def use_in_tree_miopen_cache():
    """The image ships no gfx950 MIOpen database, so on a fresh box the U-Net's first steps compile (and, for every new
    convolution shape, search) their kernels: 35 s at config 2, 5 min at configs[3].  MIOpen keeps what it built in a
    user cache; when ``svbrdf_estimation_amd/training/miopen_cache/`` exists (kernel binaries + find results written by an
    earlier run on an MI355X: ``tools/profile_train.sh`` packs them, ``tools/install_miopen_cache.sh`` unpacks them
    here; tracked since round 4 with a MANIFEST.json of file hashes, versions and provenance, because a tracked test's
    run time depends on it) and the user has not pointed MIOpen elsewhere, use it.  Purely a
    compile/search cache: the kernels are the ones a fresh box builds for itself.  Must run before the first
    convolution of the process.  Returns the directory used, or None."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_cache")
    if not os.path.isdir(os.path.join(here, "cache")):
        return None
    if os.environ.get("MIOPEN_CUSTOM_CACHE_DIR") == os.path.join(here, "cache"):
        return here                     # a parent process (the test suite) already pointed MIOpen at the in-tree cache
    if "MIOPEN_CUSTOM_CACHE_DIR" in os.environ or "MIOPEN_USER_DB_PATH" in os.environ:
        return None
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(here, "cache")
    if os.path.isdir(os.path.join(here, "db")):
        os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(here, "db")
    return here


def miopen_cache_identity(cache_dir):
    """What a result was produced with: for the in-tree cache (``use_in_tree_miopen_cache`` returned its directory) the
    sha256 of its MANIFEST.json, whether the files on disk still are the ones the manifest lists (MIOpen appends to the
    cache when it meets a new shape), and the MIOpen build it was made with; otherwise a statement that MIOpen's own
    default (or the user's) cache is in use."""
    import hashlib
    import json
    import os
    if not cache_dir:
        return {"in_tree": False, "note": "the user's / MIOpen's default cache"}
    man = os.path.join(cache_dir, "MANIFEST.json")
    if not os.path.exists(man):
        return {"in_tree": True, "manifest_sha256": None, "note": "in-tree cache without a MANIFEST.json (installed by hand)"}
    with open(man, "rb") as f:
        raw = f.read()
    spec = json.loads(raw.decode())
    changed = {}            # file -> bytes it grew by since the manifest (MIOpen appends find results and kernels), or "missing"
    for rel, want in spec.get("files", {}).items():
        path = os.path.join(cache_dir, rel)
        try:
            with open(path, "rb") as f:
                data = f.read()
            if hashlib.sha256(data).hexdigest() != want["sha256"]:
                changed[rel] = len(data) - want.get("bytes", 0)
        except OSError:
            changed[rel] = "missing"
    return {"in_tree": True, "manifest_sha256": hashlib.sha256(raw).hexdigest(), "files_match_manifest": not changed,
            "files_grown_since_manifest_bytes": changed, "miopen_build": spec.get("miopen_build")}
