"""what sits either side of the hot path in a training run -- the
Deschaintre U-Net (stock PyTorch-ROCm modules, no custom kernels), the tiled-PNG sample reader,
and a one-process-per-GPU DDP harness (RCCL all-reduce of the U-Net gradients; the rendering
loss itself shards by batch with no collective).  See train.py at the repository root."""
from . import data, models  # noqa: F401


_working_copy = {}      # in-tree cache directory -> the writable copy MIOpen was pointed at
_locks = []             # open .lock files of the copies this process holds (flock: released when the process ends)
_MAX_COPIES = 64


def _not_private(path):
    """why `path` may not be trusted as a private directory of this user (None: it may): it must be a real directory (no
    symlink), owned by this uid, without any permission for group or others"""
    import os
    import stat
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode):
        return "is not a directory (a symlink or file sits there)"
    if st.st_uid != os.getuid():
        return "is owned by uid %d, not by this user (%d)" % (st.st_uid, os.getuid())
    if st.st_mode & 0o077:
        return "is open to group/others (mode %o; need 0700)" % (st.st_mode & 0o777)
    return None


def _lock_copy(dst):
    """exclusive, non-blocking flock on <dst>/.lock, held until this process exits -> True if this process now owns the copy"""
    import fcntl
    import os
    fd = os.open(os.path.join(dst, ".lock"), os.O_RDWR | os.O_CREAT, 0o600)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
    except OSError:
        os.close(fd)
        return False
    _locks.append(fd)
    return True


def _manifest_mismatches(cache_dir):
    """-> (manifest sha256 | None, {relative path: bytes grown | "missing"}) for the files MANIFEST.json lists"""
    import hashlib
    import json
    import os
    man = os.path.join(cache_dir, "MANIFEST.json")
    if not os.path.exists(man):
        return None, {}
    with open(man, "rb") as f:
        raw = f.read()
    spec = json.loads(raw.decode())
    changed = {}
    for rel, want in spec.get("files", {}).items():
        try:
            with open(os.path.join(cache_dir, rel), "rb") as f:
                data = f.read()
            if hashlib.sha256(data).hexdigest() != want["sha256"]:
                changed[rel] = len(data) - want.get("bytes", 0)
        except OSError:
            changed[rel] = "missing"
    return hashlib.sha256(raw).hexdigest(), changed


def use_in_tree_miopen_cache(source=None, home=None):
    """The image ships no gfx950 MIOpen database, so on a fresh box the U-Net's first steps compile (and, for every new
    convolution shape, search) their kernels: 35 s at config 2, 5 min at configs[3].  MIOpen keeps what it built in a
    user cache; ``svbrdf_estimation_amd/training/miopen_cache/`` holds one (kernel binaries + find results written by an
    earlier run on an MI355X, tracked with a MANIFEST.json of file hashes, versions and provenance, because a tracked
    test's run time depends on it).  MIOpen APPENDS to its user cache whenever it meets a new shape, so it is never
    pointed at the tracked files themselves (round 4 did, and every GPU test run dirtied the work tree; a read-only
    install would fail; ranks wrote one sqlite file concurrently).  Instead:

      * the tracked files are checked against MANIFEST.json; a cache whose files no longer match is REFUSED (warning,
        MIOpen's own default cache is used): nobody runs an unreviewable binary blob that differs from the recorded one;
      * they are copied to a writable PRIVATE directory, ``<home>/miopen-<manifest sha256[:16]>/r<LOCAL_RANK>[.<slot>]``
        (home: $SVBRDF_MIOPEN_CACHE_HOME, else ~/.cache/svbrdf_amd, else ``<tmp>/svbrdf_amd-<uid>``), and
        MIOPEN_CUSTOM_CACHE_DIR / MIOPEN_USER_DB_PATH point there.  Every directory on the way is created with mode 0700
        and must be owned by this user and closed to group and others -- a directory found there that is not (another
        local user can pre-create a predictable path under /tmp and fill it with kernel binaries of their own) is refused,
        never adopted.  A copy belongs to ONE process tree at a time: the process holds an exclusive ``flock`` on its
        ``.lock`` for as long as it lives, and one that finds ``r0`` taken (another job of the same user on this node,
        a second test session) takes ``r0.1``, ``r0.2``, ...: no two jobs write one sqlite file, and the number of copies is
        bounded by the number of jobs that ever ran at once;
      * a user who has set either variable keeps their own cache.

    Purely a compile/search cache: the kernels are the ones a fresh box builds for itself.  Must run before the first
    convolution of the process.  Returns the in-tree directory whose copy is in use, or None."""
    import os
    import shutil
    import tempfile
    import warnings
    here = source or os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_cache")
    if not os.path.isdir(os.path.join(here, "cache")):
        return None
    rank_dir = "r%s" % os.environ.get("LOCAL_RANK", "0")
    if os.environ.get("SVBRDF_MIOPEN_CACHE_SOURCE") == here and os.environ.get("MIOPEN_CUSTOM_CACHE_DIR"):
        # a parent process (the test suite, a self-spawning launcher) already did this: its copy serves this process too
        # when it is the same local rank's; another rank takes a copy of its own next to it
        inherited = os.path.dirname(os.environ["MIOPEN_CUSTOM_CACHE_DIR"])
        if os.path.basename(inherited).split(".")[0] == rank_dir:
            _working_copy.setdefault(here, inherited)
            return here
        home = home or os.path.dirname(os.path.dirname(inherited))
    elif "MIOPEN_CUSTOM_CACHE_DIR" in os.environ or "MIOPEN_USER_DB_PATH" in os.environ:
        return None
    sha, changed = _manifest_mismatches(here)
    if sha is None or changed:
        warnings.warn("the in-tree MIOpen cache %s is not what its MANIFEST.json records (%s): not used" % (
            here, "no manifest" if sha is None else ", ".join(sorted(changed))))
        return None
    bases = [home or os.environ.get("SVBRDF_MIOPEN_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache", "svbrdf_amd"),
             os.path.join(tempfile.gettempdir(), "svbrdf_amd-%d" % os.getuid())]
    refused = []
    for base in bases:
        root = os.path.join(base, "miopen-%s" % sha[:16])
        try:
            for d in (base, root):
                os.makedirs(d, mode=0o700, exist_ok=True)
                why = _not_private(d)
                if why:
                    raise PermissionError("%s %s" % (d, why))
            for slot in range(_MAX_COPIES):
                dst = os.path.join(root, rank_dir if slot == 0 else "%s.%d" % (rank_dir, slot))
                if not os.path.isdir(os.path.join(dst, "cache")):
                    tmp = tempfile.mkdtemp(prefix=os.path.basename(dst) + ".tmp", dir=root)      # mkdtemp: mode 0700
                    for sub in ("cache", "db"):
                        if os.path.isdir(os.path.join(here, sub)):
                            shutil.copytree(os.path.join(here, sub), os.path.join(tmp, sub))
                    try:
                        os.rename(tmp, dst)         # atomic: a concurrent starter of the same directory wins or loses whole
                    except OSError:
                        shutil.rmtree(tmp, ignore_errors=True)
                why = _not_private(dst)
                if why:
                    raise PermissionError("%s %s" % (dst, why))
                if not os.access(os.path.join(dst, "cache"), os.W_OK):
                    raise PermissionError("%s is not writable" % os.path.join(dst, "cache"))
                if _lock_copy(dst):
                    break                           # ours for the life of this process
            else:
                raise OSError("all %d copies under %s are in use" % (_MAX_COPIES, root))
        except OSError as e:
            refused.append(str(e))
            continue
        os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(dst, "cache")
        if os.path.isdir(os.path.join(dst, "db")):
            os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(dst, "db")
        os.environ["SVBRDF_MIOPEN_CACHE_SOURCE"] = here
        _working_copy[here] = dst
        return here
    bases = ["%s" % r for r in refused] or bases
    warnings.warn("no writable directory for a copy of the in-tree MIOpen cache (tried %s): not used" % ", ".join(bases))
    return None


def miopen_cache_identity(cache_dir):
    """What a result was produced with: for the in-tree cache (``use_in_tree_miopen_cache`` returned its directory) the
    sha256 of its MANIFEST.json, whether the TRACKED files are the ones the manifest lists (they must be: MIOpen writes to
    the working copy only), where that copy is and by how much MIOpen has grown it, and the MIOpen build the cache was
    made with; otherwise a statement that MIOpen's own default (or the user's) cache is in use."""
    import json
    import os
    if not cache_dir:
        return {"in_tree": False, "note": "the user's / MIOpen's default cache"}
    sha, changed = _manifest_mismatches(cache_dir)
    if sha is None:
        return {"in_tree": True, "manifest_sha256": None, "note": "in-tree cache without a MANIFEST.json (installed by hand)"}
    with open(os.path.join(cache_dir, "MANIFEST.json")) as f:
        spec = json.load(f)
    out = {"in_tree": True, "manifest_sha256": sha, "files_match_manifest": not changed,
           "files_grown_since_manifest_bytes": changed, "miopen_build": spec.get("miopen_build")}
    work = _working_copy.get(cache_dir)
    if work:
        grown = {}
        for rel, want in spec.get("files", {}).items():
            try:
                grown[rel] = os.path.getsize(os.path.join(work, rel)) - want.get("bytes", 0)
            except OSError:
                grown[rel] = "missing"
        out["working_copy"] = work
        out["working_copy_grown_bytes"] = {k: v for k, v in grown.items() if v}
    return out
