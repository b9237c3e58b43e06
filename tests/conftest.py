import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def refmods():
    """compiled reference natives (oracle/_ref); skip if they were not built"""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (run `make -C oracle ref` where /root/reference exists)")
    return ref
