"""Builds the specialised solver library for a symbolically defined model: codegen.py emits the device
model struct, hipcc compiles csrc/usvmpc.hip with only that model's kernel instantiation (acados likewise
generates and compiles C when an AcadosOcpSolver is created).  Libraries are cached in-tree under
csrc/gen/<digest>/ so that they travel with the source tree."""
import os
import subprocess

from . import codegen

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
GEN = os.path.join(CSRC, "gen")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _dir(info, kch, soft):
    d = os.path.join(GEN, "%s_k%d_s%d" % (codegen.digest(info), kch, int(soft)))
    os.makedirs(d, exist_ok=True)
    hdr = os.path.join(d, "model_gen.hpp")
    text = codegen.emit_device_header(info)
    if not os.path.exists(hdr) or open(hdr).read() != text:
        with open(hdr, "w") as f:
            f.write(text)
    return d, hdr


def _defs(info, hdr, kch, soft):
    return ['-DUSV_GEN_MODEL_HEADER="%s"' % hdr, "-DUSV_GEN_NX=%d" % info.nx, "-DUSV_GEN_NU=%d" % info.nu,
            "-DUSV_GEN_KCH=%d" % kch, "-DUSV_GEN_SOFT=%d" % int(soft), "-DUSV_GEN_ONLY=1"]


def _stale(out, hdr):
    if not os.path.exists(out):
        return True
    srcs = [hdr] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(CSRC, "gfx950", "lanes.hpp"))
    return any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build_device_lib(info, kch, soft):
    """gfx950 library holding the generated model (model id 3). Returns its path."""
    d, hdr = _dir(info, kch, soft)
    out = os.path.join(d, "libusvmpc_gen.so")
    if _stale(out, hdr):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(CSRC, "gfx950"),
               "-I" + CSRC] + _defs(info, hdr, kch, soft) + ["-o", out, os.path.join(CSRC, "usvmpc.hip")]
        subprocess.check_call(cmd)
        # the hand-placed v_fmac_f64_dpp have one hazard the compiler does not pad inside asm: the binary is checked (dpp_check.py).  The stock
        # library is refused on a violation (the cure is a lanes::settle() at the flagged site); a USER's model must not depend on editing the
        # library's sources, so its library is rebuilt once with two wait states in front of every fused group (-DUSV_DPP_PAD: the same
        # arithmetic, a few per cent slower) and checked again
        from . import dpp_check
        n, bad = dpp_check.check_library(out)
        if bad:
            subprocess.check_call(cmd[:-3] + ["-DUSV_DPP_PAD"] + cmd[-3:])
            n, bad2 = dpp_check.check_library(out)
            if bad2:
                os.remove(out)
                raise RuntimeError("generated model: DPP hazard in the built kernels, also with padded groups:\n  " + "\n  ".join(bad2))
            open(os.path.join(d, "dpp_pad.txt"), "w").write("rebuilt with -DUSV_DPP_PAD after:\n  " + "\n  ".join(bad) + "\n")
    return out


def build_emu_lib(info, kch, soft, emu_dir):
    """CPU lane-emulator build of the same kernels with the generated model (tests only)."""
    d, hdr = _dir(info, kch, soft)
    out = os.path.join(d, "libusv_emu_gen.so")
    if _stale(out, hdr) or os.path.getmtime(os.path.join(emu_dir, "emu_driver.cpp")) > os.path.getmtime(out):
        cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + emu_dir, "-I" + CSRC] + _defs(info, hdr, kch, soft) + \
              ["-o", out, os.path.join(emu_dir, "emu_driver.cpp")]
        subprocess.check_call(cmd)
    return out
