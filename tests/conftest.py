import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __init__(self):
            self._c = {}

        def __call__(self, name):
            if name not in self._c:
                self._c[name] = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            return self._c[name]

    return G()


@pytest.fixture
def debug_lib():
    """the -DWM_DEBUG build of the kernel library for the duration of one test: every op of the test runs on it and its
    wm_debug_* switches may be flipped (the release library, which everything else loads, has none)"""
    from video_watermarking_forgery_detection_amd import _lib
    with _lib.use_debug_library() as L:
        yield L
