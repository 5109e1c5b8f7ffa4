import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ob():
    import oracle_binding

    oracle_binding.build()
    return oracle_binding


@pytest.fixture(scope="session")
def rt():
    """The product package; importing it fails loudly when the HIP library is not built."""
    import raytracer_amd

    return raytracer_amd


@pytest.fixture(scope="session")
def earth_rgb8():
    return np.load(os.path.join(GOLDEN, "earthmap_rgb8.npz"))["rgb8"]


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
