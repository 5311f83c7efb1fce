#!/usr/bin/env python3
"""Instruction mix of the innermost loops of one kernel: tools/loop_mix.py <lib.so> <kernel substring> [min loop size]"""
import re, subprocess, sys, tempfile, os, collections
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
# innermost = loops that contain no other loop of >= minsz
big = [l for l in loops if l[1] - l[0] >= minsz]
inner = [l for l in big if not any(o != l and o[0] >= l[0] and o[1] <= l[1] and (o[1]-o[0]) < (l[1]-l[0]) for o in big)]
def cls(mn, ops):
    if mn in ("v_readlane_b32", "v_writelane_b32"): return "sgpr spill (v_readlane/v_writelane)"
    if mn.startswith("v_fmac_f64_dpp"): return "fp64 fma with dpp operand"
    if mn.startswith("v_mov_b64_dpp") or mn.startswith("v_mov_b32_dpp"): return "dpp move"
    if re.match(r"v_(fma|fmac|mul|add|max|min)_f64", mn): return "fp64 arithmetic"
    if re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp|trig|cmp|cmpx|cvt).*f64", mn) or mn.startswith("v_cmp"): return "fp64 special / compare"
    if mn.startswith("v_cndmask"): return "v_cndmask_b32"
    if mn.startswith("v_mov_b64") or mn.startswith("v_mov_b32") or mn.startswith("v_accvgpr"): return "plain move"
    if mn.startswith("ds_"): return "lds"
    if mn.startswith("buffer_") or mn.startswith("global_") or mn.startswith("scratch_"): return "vmem (" + mn.split("_")[0] + ("_store" if "store" in mn else "_load") + ")"
    if mn == "s_nop": return "s_nop"
    if mn == "s_waitcnt": return "s_waitcnt"
    if mn.startswith("s_"): return "scalar"
    if mn.startswith("v_"): return "other valu"
    return "other"
for (t, i) in sorted(inner):
    c = collections.Counter(cls(mn, ops) for (_, mn, ops) in ins[t:i + 1])
    print(f"loop {ins[t][0]:#x}..{ins[i][0]:#x}: {i - t + 1} instructions")
    for k, v in c.most_common(): print(f"   {v:5d}  {k}")
