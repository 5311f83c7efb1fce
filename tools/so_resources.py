#!/usr/bin/env python3
"""Registers / spills / scratch / LDS of the kernels inside a built solver library (from the code object's metadata):
tools/so_resources.py <lib.so> [name substring]"""
import re, subprocess, sys, tempfile
L = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
subprocess.run([f"{L}/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fat.bin", lib, f"{tmp}/s"], check=True)
subprocess.run([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={tmp}/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/co.o"], check=True)
notes = subprocess.run([f"{L}/llvm-readelf", "--notes", f"{tmp}/co.o"], capture_output=True, text=True).stdout
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("usv::", "")
        print("%-64s vgpr %s spill %s scratch %s lds %s sgpr %s sgpr_spill %s" % (dem[:64], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("sgpr_count"), g("sgpr_spill_count")))
