import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def orc():
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


class _Routes:
    """Kernel-selection switches for one test (mi3d_debug_set_route: the library reads the environment only once, so tests
    switch routes through the ABI); everything is put back when the test ends."""

    def __init__(self):
        self.saved = {}

    def set(self, name, value):
        from multimodal_segmentation_project_amd import _lib
        if name not in self.saved:
            self.saved[name] = _lib.get_route(name)
        _lib.set_route(name, value)

    def reset(self, name):
        from multimodal_segmentation_project_amd import _lib
        if name in self.saved:
            _lib.set_route(name, self.saved.pop(name))

    def restore(self):
        for name in list(self.saved):
            self.reset(name)


@pytest.fixture
def routes():
    r = _Routes()
    yield r
    r.restore()
