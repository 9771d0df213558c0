import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def api():
    from raytracing_amd import api as _api
    _api.load()
    return _api


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def ctx(api):
    """One rt_ctx on cuda:0 for the whole GPU session. No fallback: fails if the device is missing."""
    c = api.Context(0)
    yield c
    c.close()


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def gamma(img):
    """`Screen::encode`'s gamma-2 (screen.rs:36-40) without the quantisation."""
    return np.sqrt(np.clip(img, 0.0, 1.0))


def rmse_gamma(a, b):
    return float(np.sqrt(((gamma(a) - gamma(b)) ** 2).mean()))
