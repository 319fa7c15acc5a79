import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


class _KgeSwitches:
    """The library's measurement switches (include/kge_amd_debug.h: kge_debug_set_switch) for the tests that compare
    kernel generations.  Until round 6 these were environment variables the library read on every call."""

    def __init__(self):
        self.touched = set()

    def set(self, name, value):
        from kge_amd import _lib
        self.touched.add(name)
        _lib.set_switch(name, int(value))

    def unset(self, name):
        from kge_amd import _lib
        self.touched.add(name)
        _lib.set_switch(name, None)

    def reset(self):
        from kge_amd import _lib
        for name in self.touched:
            _lib.set_switch(name, None)
        self.touched.clear()


@pytest.fixture
def kge_switch():
    sw = _KgeSwitches()
    try:
        yield sw
    finally:
        sw.reset()
