import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # SD_TEST_OPTIONS="name=value,..." (test runs only): start the session with non-default library switches, e.g. to run the whole
    # parity suite through a formulation that is off by default
    opts = os.environ.get("SD_TEST_OPTIONS", "")
    if opts:
        from stardist_amd.lib import _native
        for kv in opts.split(","):
            k, v = kv.split("=")
            _native.check(_native.lib().sd_set_option(k.strip().encode(), int(v)))


@pytest.fixture(scope="session")
def refmods():
    """compiled reference natives (oracle/_ref); skip if they were not built"""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (run `make -C oracle ref` where /root/reference exists)")
    return ref
