"""Builds tests/emu/libusv_emu.so (TEST-ONLY): the kernel bodies of mpc_collisionavoidance_amd/csrc compiled by the host compiler against the
lane emulator.  emu_driver.cpp is compiled as six translation units in parallel (its "Build parts"): ~2.5 min instead of 7."""
import os
import subprocess

EMU = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(EMU))
CSRC = os.path.join(ROOT, "mpc_collisionavoidance_amd", "csrc")
OUT = os.path.join(EMU, "libusv_emu.so")
PARTS = 6


def sources():
    return [os.path.join(EMU, "emu_driver.cpp"), os.path.join(EMU, "lanes.hpp"), os.path.join(ROOT, "include", "usvmpc.h")] + \
           [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]


def stale():
    return not os.path.exists(OUT) or any(os.path.getmtime(s) > os.path.getmtime(OUT) for s in sources())


def build(force=False):
    if not (force or stale()):
        return OUT
    obj = os.path.join(EMU, "build")
    os.makedirs(obj, exist_ok=True)
    flags = ["g++", "-O1", "-std=c++17", "-fPIC", "-I" + EMU, "-I" + CSRC]
    objs = [os.path.join(obj, "p%d.o" % p) for p in range(PARTS)]
    procs = [subprocess.Popen(flags + ["-DEMU_PART=%d" % p, "-c", "-o", objs[p], os.path.join(EMU, "emu_driver.cpp")]) for p in range(PARTS)]
    if any(p.wait() != 0 for p in procs):
        raise RuntimeError("g++ failed for the lane emulator")
    subprocess.check_call(["g++", "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv)
