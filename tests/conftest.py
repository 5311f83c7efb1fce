import os
import sys

import pytest
import torch  # noqa: F401  - before libusvmpc.so is loaded, so that both use one HIP runtime (see _capi.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding
