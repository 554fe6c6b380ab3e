import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs them on a GPU-less box they must
    # fail loudly rather than skip (a silent skip would read as "parity green").
    pass


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
