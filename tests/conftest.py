import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def host():
    from rustracer_amd import host as _host
    _host.build()
    return _host


@pytest.fixture(scope="session")
def gpu_host(host):
    if not host.device_available():
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (no CPU fallback exists)")
    return host
