import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


# the in-process training tests run MIOpen convolutions: use the in-tree kernel cache when one has been installed
from svbrdf_estimation_amd.training import use_in_tree_miopen_cache  # noqa: E402

use_in_tree_miopen_cache()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """the allowance ledger of this session (tests/tolerances.py): every use of a tolerance widening / tie exclusion, as
    printed by the tests, in one file -- gpurun_out/ is what comes back from the GPU box"""
    try:
        import tolerances
        if not tolerances.ALLOWANCES_USED:
            return
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        import torch
        where = torch.cuda.get_device_name(0) if torch.cuda.is_available() else "CPU only (oracle vs fixtures)"
        with open(os.path.join(out, "tolerance_uses.txt"), "w") as f:
            f.write("# allowance ledger of one pytest session (tests/tolerances.py); device: %s; exit status %s\n" % (where, exitstatus))
            f.write("# what                                                 kind                     used of      total (cap)\n")
            f.write("\n".join(tolerances.ledger_lines()) + "\n")
    except Exception as e:      # a reporting aid must never turn a green run red
        print("[tolerance] ledger not written: %r" % (e,))


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs them on a GPU-less box they must
    # fail loudly rather than skip (a silent skip would read as "parity green").
    pass


@pytest.fixture(scope="session", autouse=True)
def native_artifacts():
    """The suites need the in-tree native builds (HIP library, host extension, C oracle).  They normally
    exist already (`__graft_entry__.build()`); on a fresh checkout they are built here once -- hipcc
    cross-compiles gfx950 without a GPU."""
    import subprocess
    lib = os.path.join(ROOT, "svbrdf_estimation_amd", "lib", "libsvbrdf_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "svbrdf_estimation_amd", "csrc")])
    from svbrdf_estimation_amd import _hostext
    if not os.path.exists(_hostext._SO) and not os.environ.get("SVBRDF_NO_HOST_EXT"):
        _hostext.build()
    from oracle import c_oracle
    c_oracle.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    gdir = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return np.load(os.path.join(gdir, name))
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle
