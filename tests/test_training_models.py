"""CPU: row f4 -- the re-stated Deschaintre network.  Parameter counts are the reference's (SURVEY.md
section 2 probe), and where the reference is mounted (the build container) the forward pass is
compared with the reference's own models after converting its state dict."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference/development/multiImage_pytorch"


def test_parameter_counts_match_the_reference():
    from svbrdf_estimation_amd.training import models
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(models.SingleViewModel(use_coords=False)) == 79985621
    assert count(models.MultiViewModel(use_coords=False)) == 80262042
    assert count(models.SingleViewModel(use_coords=True)) == 79985621 + 2 * (64 * 16 + 128)


def _load_reference_models():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.modules.setdefault("pyredner", types.ModuleType("pyredner"))
    saved = {k: sys.modules.get(k) for k in ("utils", "models")}
    sys.path.insert(0, REF)
    try:
        for k in ("utils", "models"):
            sys.modules.pop(k, None)
        import models as ref_models
        return ref_models
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only mounted in the build container")
@pytest.mark.parametrize("use_coords", [True, False])
def test_forward_equals_reference_models(use_coords):
    from svbrdf_estimation_amd.training import models
    ref_models = _load_reference_models()
    torch.manual_seed(0)
    ref = ref_models.SingleViewModel(use_coords=use_coords).eval()
    mine = models.SingleViewModel(use_coords=use_coords).eval()
    mine.load_state_dict(models.convert_reference_state_dict(ref.state_dict()))
    x = torch.rand(1, 3, 256, 256)
    with torch.no_grad():
        a, b = ref(x), mine(x)
    assert a.shape == b.shape == (1, 12, 256, 256)
    np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=1e-5, atol=1e-6)
    enc = models.SingleViewModel(use_coords=use_coords, decode=False).eval()
    enc.load_state_dict(mine.state_dict())
    with torch.no_grad():
        assert enc(x).shape == (1, 9, 256, 256)
    if use_coords:       # multi-view: N photos, max-pooled; the reference loops over photos, this batches them
        torch.manual_seed(1)
        ref_mv = ref_models.MultiViewModel(use_coords=True).eval()
        mine_mv = models.MultiViewModel(use_coords=True).eval()
        mine_mv.load_state_dict(models.convert_reference_state_dict(ref_mv.state_dict()))
        xs = torch.rand(1, 2, 3, 256, 256)
        with torch.no_grad():
            # batched generator pass vs the reference's per-photo loop: different conv blocking, fp32 noise
            np.testing.assert_allclose(mine_mv(xs).numpy(), ref_mv(xs).numpy(), rtol=1e-4, atol=3e-5)


def test_tiled_png_round_trip(tmp_path):
    from svbrdf_estimation_amd.training import data
    torch.manual_seed(3)
    H, n = 32, 2
    photos = torch.rand(n, 3, H, H)
    normals = torch.nn.functional.normalize(torch.randn(3, H, H) * 0.3 + torch.tensor([0, 0, 1.0]).view(3, 1, 1), dim=0)
    svbrdf = torch.cat((normals, torch.rand(9, H, H)), dim=0)
    path = str(tmp_path / "sample.png")
    data.write_tiled_png(path, photos, svbrdf)
    p2, s2 = data.read_tiled_png(path, n)
    assert p2.shape == (n, 3, H, H) and s2.shape == (12, H, H)
    assert (p2 - photos).abs().max() <= 0.5 / 255 + 1e-6
    assert (s2[3:] - svbrdf[3:]).abs().max() <= 0.5 / 255 + 1e-6
    assert (s2[:3] - svbrdf[:3]).abs().max() <= 1.0 / 255 + 1e-6          # normals stored in [0,1]
    ds = data.TiledPngDataset(str(tmp_path), image_size=16, image_count=n, used_image_count=1)
    item = ds[0]
    assert item["inputs"].shape == (1, 3, 16, 16) and item["svbrdf"].shape == (12, 16, 16)
    assert torch.allclose(item["inputs"][0], p2[1, :, :16, :16] ** 2.2)  # last photo, gamma-decoded
    with pytest.raises(ValueError):
        data.read_tiled_png(path, 3)
    syn = data.SyntheticSvbrdfDataset(4, image_size=8, seed=1)
    assert len(syn) == 4 and syn[2]["svbrdf"].shape == (12, 8, 8) and torch.equal(syn[2]["svbrdf"], syn[2]["svbrdf"])


def test_in_tree_miopen_cache_is_what_its_manifest_says():
    """tests/test_gpu_parity.py's at-size configs[3] harness test takes 1 s with the in-tree MIOpen user cache and
    ~5 min (compiling) without: the cache is tracked, and its MANIFEST.json (file hashes, MIOpen build, provenance:
    tools/miopen_cache_manifest.py) must describe the files that are there; train.py prints the manifest's sha256."""
    import hashlib
    import json
    from svbrdf_estimation_amd import training
    here = os.path.join(os.path.dirname(os.path.abspath(training.__file__)), "miopen_cache")
    with open(os.path.join(here, "MANIFEST.json")) as f:
        man = json.load(f)
    assert man["arch"] == "gfx950" and man["produced_by"] and man["miopen_build"] and len(man["files"]) >= 2
    on_disk = sorted(os.path.relpath(os.path.join(r, n), here) for r, _, ns in os.walk(here) for n in ns if n != "MANIFEST.json")
    assert on_disk == sorted(man["files"])
    for rel, want in man["files"].items():
        with open(os.path.join(here, rel), "rb") as f:
            raw = f.read()
        assert len(raw) == want["bytes"] and hashlib.sha256(raw).hexdigest() == want["sha256"], rel
    ident = training.miopen_cache_identity(here)
    assert ident["in_tree"] and ident["files_match_manifest"] and len(ident["manifest_sha256"]) == 64
    assert training.miopen_cache_identity(None) == {"in_tree": False, "note": "the user's / MIOpen's default cache"}


def test_miopen_cache_is_used_through_a_writable_copy_and_refused_when_tampered(tmp_path, monkeypatch):
    """MIOpen appends to its user cache: it must never be pointed at the tracked files (round 4: every GPU test run
    dirtied the work tree).  The in-tree cache is verified against its manifest, copied to a per-user directory (one per
    local rank) and used from there; a cache whose files differ from the manifest is refused."""
    import shutil
    import warnings
    from svbrdf_estimation_amd import training
    here = os.path.join(os.path.dirname(os.path.abspath(training.__file__)), "miopen_cache")
    for var in ("MIOPEN_CUSTOM_CACHE_DIR", "MIOPEN_USER_DB_PATH", "SVBRDF_MIOPEN_CACHE_SOURCE", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(training, "_working_copy", {})
    home = str(tmp_path / "home")
    assert training.use_in_tree_miopen_cache(home=home) == here
    cache_dir = os.environ["MIOPEN_CUSTOM_CACHE_DIR"]
    assert cache_dir.startswith(home) and not cache_dir.startswith(here) and cache_dir.endswith(os.path.join("r0", "cache"))
    assert os.environ["MIOPEN_USER_DB_PATH"].startswith(home) and os.environ["SVBRDF_MIOPEN_CACHE_SOURCE"] == here
    assert sorted(os.listdir(cache_dir)) == sorted(os.listdir(os.path.join(here, "cache")))
    with open(os.path.join(cache_dir, os.listdir(cache_dir)[0]), "ab") as f:       # MIOpen appends ...
        f.write(b"x" * 10)
    ident = training.miopen_cache_identity(here)
    assert ident["files_match_manifest"] and ident["working_copy"] == os.path.dirname(cache_dir)    # ... the tracked files stay
    assert list(ident["working_copy_grown_bytes"].values()) == [10]
    # a child process of this one (same local rank) reuses the copy; another local rank takes its own next to it
    monkeypatch.setattr(training, "_working_copy", {})
    assert training.use_in_tree_miopen_cache() == here and os.environ["MIOPEN_CUSTOM_CACHE_DIR"] == cache_dir
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert training.use_in_tree_miopen_cache() == here
    assert os.environ["MIOPEN_CUSTOM_CACHE_DIR"] == os.path.join(os.path.dirname(os.path.dirname(cache_dir)), "r3", "cache")
    # another job of the same user on this node (nothing inherited, LOCAL_RANK unset again): r0 is held -- flock -- by the
    # first, so it gets a copy of its own next to it instead of writing the same sqlite file
    for var in ("MIOPEN_CUSTOM_CACHE_DIR", "MIOPEN_USER_DB_PATH", "SVBRDF_MIOPEN_CACHE_SOURCE", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(training, "_working_copy", {})
    assert training.use_in_tree_miopen_cache(home=home) == here
    assert os.environ["MIOPEN_CUSTOM_CACHE_DIR"] == os.path.join(os.path.dirname(os.path.dirname(cache_dir)), "r0.1", "cache")
    # every directory on the way is private: created 0700, and a directory somebody else could have prepared (here: one
    # open to group/others; a foreign owner cannot be staged without root) is refused, not adopted
    import stat
    for d in (home, os.path.dirname(os.path.dirname(cache_dir)), os.path.dirname(cache_dir)):
        assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700, d
    for var in ("MIOPEN_CUSTOM_CACHE_DIR", "MIOPEN_USER_DB_PATH", "SVBRDF_MIOPEN_CACHE_SOURCE"):
        monkeypatch.delenv(var, raising=False)
    planted = str(tmp_path / "planted")
    os.makedirs(planted, mode=0o755)
    os.chmod(planted, 0o755)
    (tmp_path / "a-file").write_text("not a directory")
    monkeypatch.setattr("tempfile.gettempdir", lambda: str(tmp_path / "a-file"))      # no second base to fall back on
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert training.use_in_tree_miopen_cache(home=planted) is None
    assert w and "open to group/others" in str(w[0].message) and "MIOPEN_CUSTOM_CACHE_DIR" not in os.environ
    assert os.listdir(planted) == []
    monkeypatch.undo()
    for var in ("MIOPEN_CUSTOM_CACHE_DIR", "MIOPEN_USER_DB_PATH", "SVBRDF_MIOPEN_CACHE_SOURCE", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    # the user's own setting wins
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert training.use_in_tree_miopen_cache(home=home) is None and "MIOPEN_CUSTOM_CACHE_DIR" not in os.environ
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    # tampered source: refused, MIOpen left alone
    src = str(tmp_path / "src")
    shutil.copytree(here, src)
    victim = os.path.join(src, "cache", os.listdir(os.path.join(src, "cache"))[0])
    with open(victim, "r+b") as f:
        f.seek(100)
        b = f.read(1)
        f.seek(100)
        f.write(bytes([b[0] ^ 1]))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert training.use_in_tree_miopen_cache(source=src, home=home) is None
    assert w and "not used" in str(w[0].message) and "MIOPEN_CUSTOM_CACHE_DIR" not in os.environ
