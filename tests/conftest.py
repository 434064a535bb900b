import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".npy"):
        return np.load(path)
    return dict(np.load(path))


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc.Oracle(long_double=False)


@pytest.fixture(scope="session")
def oracle_ld():
    import oracle as orc
    orc.build()
    return orc.Oracle(long_double=True)
