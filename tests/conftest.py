import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "2023-compact-image-compression_amd")
for p in (ROOT, PKG, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(HERE, "golden", "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")
