import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from oracle.io import load_npz

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_npz(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return get
