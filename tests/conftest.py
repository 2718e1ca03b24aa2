"""pytest configuration: markers + shared fixtures.

`-m "not gpu"`: oracle vs the golden vectors, host logic, C-ABI symbol checks (no GPU needed).
`-m gpu`      : HIP path through the C ABI vs the oracle and the golden fixtures (needs an MI355X).
Nothing here reads /root/reference: fixtures were cut once by tests/golden/make_golden.js.
"""
import os
import sys

import pytest

try:   # one HIP runtime per process: torch's bundled copy must be loaded before libcjs_hip.so (see package __init__)
    import torch  # noqa: F401
except Exception:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU cases")


@pytest.fixture(scope="session")
def oracle():
    import support
    return support.Oracle()


@pytest.fixture(scope="session")
def hip():
    import support
    return support.HipLib()
