#!/usr/bin/env python3
"""Static per-source-line instruction counts of the inner loops of one kernel (library built with -gline-tables-only):
tools/line_profile.py <lib.so> <kernel substring> [min loop size] [top n]"""
import re, subprocess, sys, tempfile, collections
L = "/opt/rocm/lib/llvm/bin"
lib, pat = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 300
topn = int(sys.argv[4]) if len(sys.argv) > 4 else 25
tmp = tempfile.mkdtemp()
subprocess.run([f"{L}/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fat.bin", lib, f"{tmp}/s"], check=True)
subprocess.run([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={tmp}/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/co.o"], check=True)
dis = subprocess.run([f"{L}/llvm-objdump", "-d", "-l", "--no-show-raw-insn", f"{tmp}/co.o"], capture_output=True, text=True).stdout.splitlines()
on = False; ins = []; cur = "?"
for ln in dis:
    m = re.match(r"^[0-9a-f]+ <([^>]*)>:", ln)
    if m:
        if not m.group(1).startswith("L"): on = pat in m.group(1)
        continue
    m = re.match(r"^; (\S+):(\d+)", ln)
    if m:
        cur = m.group(1).split("/")[-1] + ":" + m.group(2); continue
    if on:
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2), cur))
addr = {a: i for i, (a, _, _, _) in enumerate(ins)}
loops = []
for i, (a, mn, ops, _) in enumerate(ins):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        m = re.match(r"(\d+)", ops.strip())
        if m:
            off = int(m.group(1)); off = off - 0x10000 if off >= 0x8000 else off
            t = addr.get(a + 4 + 4 * off)
            if t is not None and t < i: loops.append((t, i))
big = [l for l in loops if l[1] - l[0] >= minsz]
inner = [l for l in big if not any(o != l and o[0] >= l[0] and o[1] <= l[1] and (o[1]-o[0]) < (l[1]-l[0]) for o in big)]
for (t, e) in sorted(inner):
    c = collections.Counter(); cm = collections.defaultdict(collections.Counter)
    for (_, mn, ops, ln) in ins[t:e + 1]:
        c[ln] += 1; cm[ln][mn.split("_e")[0] if mn.startswith("v_cndmask") else mn] += 1
    print(f"loop {ins[t][0]:#x}..{ins[e][0]:#x}: {e - t + 1} instructions")
    for ln, n in c.most_common(topn):
        print(f"   {n:5d}  {ln:28s} " + ", ".join(f"{k} {v}" for k, v in cm[ln].most_common(4)))
