#!/usr/bin/env python3
"""Vector-memory instructions and vmcnt waits of the inner loops of a kernel, in program order (is a wait sitting between the
prefetch of the next stage and the end of the stage?): tools/loop_vmem.py <lib.so> <kernel substring> [min loop size]"""
import re, subprocess, sys, tempfile
L = "/opt/rocm/lib/llvm/bin"
lib, pat = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 300
tmp = tempfile.mkdtemp()
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from mpc_collisionavoidance_amd import dpp_check   # (the library holds one offload bundle per translation unit: all of them)
cos, _tmp = dpp_check.code_objects(lib)
dis = sum((subprocess.run([f"{L}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout.splitlines() for co in cos), [])
on = False; ins = []
for ln in dis:
    m = re.match(r"^[0-9a-f]+ <([^>]*)>:", ln)
    if m:
        if not m.group(1).startswith("L"): on = pat in m.group(1)
        continue
    if on:
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr = {a: i for i, (a, _, _) in enumerate(ins)}
loops = []
for i, (a, mn, ops) in enumerate(ins):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        m = re.match(r"(\d+)", ops.strip())
        if m:
            off = int(m.group(1)); off = off - 0x10000 if off >= 0x8000 else off
            t = addr.get(a + 4 + 4 * off)
            if t is not None and t < i: loops.append((t, i))
big = [l for l in loops if l[1] - l[0] >= minsz]
inner = [l for l in big if not any(o != l and o[0] >= l[0] and o[1] <= l[1] and (o[1]-o[0]) < (l[1]-l[0]) for o in big)]
for (t, e) in sorted(inner):
    print(f"loop {ins[t][0]:#x}..{ins[e][0]:#x}: {e - t + 1} instructions")
    run = None
    for i in range(t, e + 1):
        a, mn, ops = ins[i]
        v = mn.startswith(("buffer_", "global_", "scratch_", "flat_"))
        if v:
            kind = ("store" if "store" in mn else "load") + (" (scratch)" if mn.startswith("scratch") else "")
            if run and run[0] == kind and i - run[2] <= 12: run[1] += 1; run[2] = i
            else:
                if run: print(f"   @{run[3]-t:5d}  {run[1]:2d} x {run[0]}")
                run = [kind, 1, i, i]
        elif mn == "s_waitcnt" and "vmcnt" in ops:
            if run: print(f"   @{run[3]-t:5d}  {run[1]:2d} x {run[0]}"); run = None
            print(f"   @{i-t:5d}  s_waitcnt {ops}")
    if run: print(f"   @{run[3]-t:5d}  {run[1]:2d} x {run[0]}")
