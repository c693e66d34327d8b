import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import native
    native.lib()
    return native


@pytest.fixture(scope="session")
def hip():
    """The loaded HIP library on a GPU box; fails (never skips) if it is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    from cloudaae_amd import _lib
    _lib.lib()
    return _lib


@pytest.fixture
def knobs(hip):
    """knobs(name, value): set a development knob of the HIP library for this test (cloudaae_set_knob; value None =
    unset); every knob touched is unset again afterwards."""
    touched = []

    def set_(name, value):
        touched.append(name)
        hip.set_knob(name, value)
    yield set_
    for name in touched:
        hip.set_knob(name, None)
