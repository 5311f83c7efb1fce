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
    EMU = os.path.join(ROOT, "tests", "emu")
    CSRC = os.path.join(ROOT, "mpc_collisionavoidance_amd", "csrc")
    out = os.path.join(EMU, "libusv_emu.so")
    srcs = [os.path.join(EMU, "emu_driver.cpp"), os.path.join(EMU, "lanes.hpp")] + \
           [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + EMU, "-I" + CSRC, "-o", out,
                               os.path.join(EMU, "emu_driver.cpp")])
    lib = C.CDLL(out)
    dp, ip = _capi._dp, _capi._ip
    lib.usv_emu_solve.argtypes = [C.POINTER(_capi.Desc)] + [dp] * 10 + [ip] * 3 + [dp] * 4
    lib.usv_emu_sqp.argtypes = [C.POINTER(_capi.Desc)] + [dp] * 10 + [ip] * 3 + [dp] + [ip] + [dp]
    return lib
