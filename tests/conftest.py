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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the reference's own warning on lego-scale points (utils/xyz.py:8-9), reproduced lazily: expected wherever a test renders
    # a camera at radius 4; tests/test_gpu_boundary.py::test_range_warning_is_the_references looks at it on purpose
    config.addinivalue_line("filterwarnings", "ignore:input not in range -1,1:UserWarning")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    import nerf_oracle
    return nerf_oracle


@pytest.fixture(scope="session")
def synthetic():
    from nerf_simple_amd.utils import synthetic
    return synthetic
