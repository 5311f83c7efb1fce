"""The static hazard check of hand-placed DPP instructions (mpc_collisionavoidance_amd/dpp_check.py).

lanes::fma_bc places v_fmac_f64_dpp through inline asm; the compiler does not pad the one hazard that instruction has (a VALU
write of its DPP source within the two preceding wait states: tools/micro/dpp_hazard.hip measured it on the part), so every
library the build produces is disassembled and checked.  Here: the in-tree library is clean and actually contains the fused
instructions, and the checker does flag a kernel written to violate the rule (also across a branch)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_collisionavoidance_amd import dpp_check  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"


def test_in_tree_library_is_clean():
    import __graft_entry__ as g
    lib = g.build_hip()
    n, bad = dpp_check.check_library(lib)
    assert n > 1000, "the fused DPP FMAs are not in the built library (%d found)" % n
    assert bad == []


BAD = r"""
#include <hip/hip_runtime.h>
__global__ void straight(double *p)
{
    double acc = p[threadIdx.x], b = p[64 + threadIdx.x], a = 2.0;
    asm volatile("v_add_f64 %1, %1, %1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(b) : "v"(a));
    p[threadIdx.x] = acc;
}
__global__ void padded(double *p)
{
    double acc = p[threadIdx.x], b = p[64 + threadIdx.x], a = 2.0;
    asm volatile("v_add_f64 %1, %1, %1\n s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(b) : "v"(a));
    p[threadIdx.x] = acc;
}
__global__ void across_branch(double *p, int n)
{
    double acc = p[threadIdx.x], b = p[64 + threadIdx.x], a = 2.0;
    asm volatile("s_cmp_eq_u32 %3, 0\n s_cbranch_scc1 1f\n v_add_f64 %1, %1, %1\n 1:\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc), "+v"(b) : "v"(a), "s"(n) : "scc");
    p[threadIdx.x] = acc;
}
"""


def test_checker_flags_a_violation(tmp_path):
    src = tmp_path / "bad.hip"
    src.write_text(BAD)
    out = tmp_path / "bad.co"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "--cuda-device-only", "-c", "-o", str(out), str(src)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("hipcc could not build the probe: " + r.stderr[-300:])
    n, bad = dpp_check.check_library(str(out))
    assert n == 3
    assert any("straight" in b for b in bad)
    assert any("across_branch" in b for b in bad)
    assert not any("padded" in b for b in bad)


def test_checker_refuses_a_fused_build_in_which_it_finds_nothing(tmp_path):
    """A fused library in which no v_fmac_f64_dpp is recognised means the disassembly was not understood: a violation, not
    a pass (an unfused build says so with expect_fused=False)."""
    src = tmp_path / "plain.hip"
    src.write_text('#include <hip/hip_runtime.h>\n__global__ void plain(double *p) { p[threadIdx.x] *= 2.0; }\n')
    out = tmp_path / "plain.co"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "--cuda-device-only", "-c", "-o", str(out), str(src)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("hipcc could not build the probe: " + r.stderr[-300:])
    n, bad = dpp_check.check_library(str(out))
    assert n == 0 and any("nothing was checked" in b for b in bad)
    n, bad = dpp_check.check_library(str(out), expect_fused=False)
    assert n == 0 and not bad


EXEC_BAD = r"""
#include <hip/hip_runtime.h>
__global__ void exec_write(double *p)
{
    double acc = p[threadIdx.x], b = p[64 + threadIdx.x], a = 2.0;
    asm volatile("s_mov_b64 s[10:11], exec\n v_cmpx_gt_f64 vcc, %2, %1\n s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_mov_b64 exec, s[10:11]"
                 : "+v"(acc) : "v"(b), "v"(a) : "s10", "s11", "vcc");
    p[threadIdx.x] = acc;
}
__global__ void exec_write_padded(double *p)
{
    double acc = p[threadIdx.x], b = p[64 + threadIdx.x], a = 2.0;
    asm volatile("s_mov_b64 s[10:11], exec\n v_cmpx_gt_f64 vcc, %2, %1\n s_nop 4\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_mov_b64 exec, s[10:11]"
                 : "+v"(acc) : "v"(b), "v"(a) : "s10", "s11", "vcc");
    p[threadIdx.x] = acc;
}
"""


def test_checker_flags_a_valu_write_of_exec_in_front_of_dpp(tmp_path):
    """Second hazard of the hand-placed instruction: a VALU write of EXEC needs 5 wait states before a DPP instruction."""
    src = tmp_path / "execbad.hip"
    src.write_text(EXEC_BAD)
    out = tmp_path / "execbad.co"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "--cuda-device-only", "-c", "-o", str(out), str(src)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("hipcc could not build the probe: " + r.stderr[-300:])
    n, bad = dpp_check.check_library(str(out))
    assert n == 2
    assert any("exec_write" in b and "EXEC" in b and "padded" not in b for b in bad), bad
    assert not any("exec_write_padded" in b for b in bad), bad


PAD_PROBE = r"""
#include "lanes.hpp"
// a VALU write of the DPP source right in front of the fused group: a hazard as written - and none once every group is padded
__global__ void probe(double *p)
{
    double acc = p[threadIdx.x], b = p[64 + threadIdx.x], a = 2.0;
    asm volatile("v_add_f64 %0, %0, %0" : "+v"(b));
    lanes::fma_bc2<1, 2>(acc, b, a, b, a);
    p[threadIdx.x] = acc;
}
"""


def test_padded_groups_cure_a_hazard_in_a_generated_build(tmp_path):
    """genbuild.py rebuilds a user's model library with -DUSV_DPP_PAD when its kernels trip the check (arbitrary-model codegen must
    not depend on editing lanes::settle() into the library's sources): two wait states in front of every fused group."""
    src = tmp_path / "pad.hip"
    src.write_text(PAD_PROBE)
    inc = "-I" + os.path.join(ROOT, "mpc_collisionavoidance_amd", "csrc", "gfx950")
    res = {}
    for tag, defs in (("plain", []), ("padded", ["-DUSV_DPP_PAD"])):
        out = tmp_path / (tag + ".co")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "--cuda-device-only", inc] + defs + ["-c", "-o", str(out), str(src)],
                           capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("hipcc could not build the probe: " + r.stderr[-300:])
        res[tag] = dpp_check.check_library(str(out))
    assert res["plain"][0] == 2 and res["padded"][0] == 2
    assert res["padded"][1] == []
    # (as written the compiler may or may not leave two independent instructions between the write and the group: when it does not, the
    # check must say so - either way the padded build is clean)
    assert all("probe" in b for b in res["plain"][1])
