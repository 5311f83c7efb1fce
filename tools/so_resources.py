#!/usr/bin/env python3
"""Registers / spills / scratch / LDS of the kernels inside a built solver library (from the code object's metadata):
tools/so_resources.py <lib.so> [name substring]"""
import re, subprocess, sys, tempfile
L = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from mpc_collisionavoidance_amd import dpp_check   # (the library holds one offload bundle per translation unit: all of them)
cos, _tmp = dpp_check.code_objects(lib)
notes = "".join(subprocess.run([f"{L}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout for co in cos)
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("usv::", "")
        print("%-64s vgpr %s spill %s scratch %s lds %s sgpr %s sgpr_spill %s" % (dem[:64], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("sgpr_count"), g("sgpr_spill_count")))
