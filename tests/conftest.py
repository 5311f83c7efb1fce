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


@pytest.fixture(scope="session")
def emu():
    """The unmodified kernel bodies compiled for the CPU lane emulator (tests/emu); rebuilt when stale."""
    import ctypes as C
    import subprocess
    from mpc_collisionavoidance_amd import _capi
    from tests.emu import build_emu
    lib = C.CDLL(build_emu.build())
    dp, ip = _capi._dp, _capi._ip
    lib.usv_emu_solve.argtypes = [C.POINTER(_capi.Desc)] + [dp] * 10 + [ip] * 3 + [dp] * 4
    lib.usv_emu_sqp.argtypes = [C.POINTER(_capi.Desc)] + [dp] * 10 + [ip] * 3 + [dp] + [ip] + [dp]
    return lib
