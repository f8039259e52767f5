"""Test configuration.

* ``-m "not gpu"``: oracle pinning, host logic, C-ABI symbol checks (no GPU needed).
* ``-m gpu``: parity of the HIP path (through the C ABI / the ``stereomatch`` package)
  against the CPU oracle and the committed golden fixtures; needs one MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stereo-matching_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def device():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc
