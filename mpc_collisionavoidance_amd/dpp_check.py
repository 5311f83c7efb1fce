#!/usr/bin/env python3
"""Static check of a gfx950 code object for the one hazard hand-placed DPP instructions can hit.

gfx950 needs 2 wait states between a VALU write of a VGPR and a DPP instruction that reads it as its DPP source (src0); the
compiler pads the DPP moves it emits itself, but not the `v_fmac_f64_dpp` that lanes::fma_bc places through inline asm.  This
script disassembles every kernel of a shared library (or a bare code object) and, for each such instruction, looks at the two
instructions issued before it on every path the disassembly shows: straight-line predecessors, and - when the instruction sits
within two slots of a label - the instructions before every branch to that label; calls of non-inlined device functions
(s_swappc_b64 / s_setpc_b64 s[30:31]) are followed conservatively (every call site in front of every callee, every callee's return
in front of every call's successor).  Anything else it cannot follow (computed branches) counts as a violation.  Exit status 1
and a listing when a violation exists.

usage: python -m mpc_collisionavoidance_amd.dpp_check <lib.so | code object> [...]   (tools/check_dpp_hazard.py wraps this)
`check_library(path)` is what __graft_entry__.build() and genbuild.build_device_lib() call on every library they produce.
"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"
DPP_ASM = ("v_fmac_f64_dpp",)          # mnemonics placed by hand
WAIT_STATES = 2                         # VALU write of the DPP source VGPR -> DPP read
EXEC_WAIT_STATES = 5                    # VALU write of EXEC (v_cmpx*, v_readlane/writelane never; any v_* naming exec as destination) -> DPP
# VALU instructions that write a SECOND register beside operand 0 (the swap's other operand): both are destinations
TWO_DEST = ("v_swap_b32", "v_swap_b16", "v_swaprel_b32")


def writes_exec(mn, ops):
    """a VALU instruction that writes the EXEC mask (only those count for the 5-wait-state DPP rule; SALU writes need none extra)"""
    if not mn.startswith("v_"):
        return False
    if mn.startswith("v_cmpx"):
        return True
    first = ops.split(",")[0].strip() if ops else ""
    return first in ("exec", "exec_lo", "exec_hi")


def code_objects(path):
    """gfx950 code objects inside `path`: a host shared library with an offload bundle, a bare bundle, or already a code object."""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    head = open(path, "rb").read(64)
    if not head.startswith(magic):
        out = subprocess.run([f"{LLVM}/llvm-readelf", "-h", path], capture_output=True, text=True).stdout
        if "AMDGPU" in out or "AMD GPU" in out:
            return [path], None
    tmp = tempfile.mkdtemp(prefix="dppchk_")
    if head.startswith(magic):
        data = open(path, "rb").read()
    else:   # the bundle sits in section .hip_fatbin of the host library
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(tmp, "stripped")], check=True)
        data = open(fat, "rb").read()
    res = []
    # one or more concatenated bundles: let clang-offload-bundler list and extract each
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for n, st in enumerate(starts):
        en = starts[n + 1] if n + 1 < len(starts) else len(data)
        part = os.path.join(tmp, f"bundle{n}")
        open(part, "wb").write(data[st:en])
        lst = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={part}"], capture_output=True, text=True).stdout.split()
        for t in lst:
            if "gfx950" not in t:
                continue
            co = os.path.join(tmp, f"co{n}_{len(res)}.o")
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--targets={t}", f"--output={co}"], check=True)
            res.append(co)
    return res, tmp


INSTR = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
LABEL = re.compile(r"^([0-9a-f]+) <([^>]+)>:")


def regs(tok):
    """VGPR numbers named by one operand token (v7, v[6:7], -v[6:7], |v3|)."""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    if m:
        return {int(m.group(1))}
    return set()


def is_valu(mn):
    return mn.startswith("v_") and not mn.startswith("v_nop")


def wait_states(mn, ops):
    if mn == "s_nop":
        return int(ops.split()[0], 0) + 1
    return 1


def check(co):
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout.splitlines()
    # parse into a flat list with labels
    prog = []           # (addr, mnemonic, operands)
    label_at = {}       # name -> index into prog
    func = None
    funcs = []
    for ln in dis:
        m = LABEL.match(ln)
        if m:
            label_at[m.group(2)] = len(prog)
            if not m.group(2).startswith("L"):
                func = m.group(2)
            continue
        m = INSTR.match(ln)
        if m:
            prog.append((int(m.group(3), 16), m.group(1), m.group(2)))
            funcs.append(func)
    addr_index = {a: i for i, (a, _, _) in enumerate(prog)}
    # branch targets: llvm-objdump prints "s_cbranch_xxx L123" style? it prints the label name <Lnn> in the comment or operand
    branches_to = {}    # target index -> [source index]
    unknown_branch = False
    for i, (a, mn, ops) in enumerate(prog):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"<([^>+]+)(\+0x[0-9a-f]+)?>", ops)
            tgt = None
            if m and m.group(1) in label_at:
                tgt = label_at[m.group(1)] if not m.group(2) else addr_index.get(prog[label_at[m.group(1)]][0] + int(m.group(2), 16))
            else:
                m2 = re.match(r"(\d+)$", ops.strip())
                if m2:   # raw simm16: target = addr + 4 + simm16*4
                    off = int(m2.group(1))
                    if off >= 0x8000:
                        off -= 0x10000
                    tgt = addr_index.get(a + 4 + 4 * off)
            if tgt is None:
                unknown_branch = True
            else:
                branches_to.setdefault(tgt, []).append(i)
    # Calls: hipcc emits `s_swappc_b64` for a call of a non-inlined device function and `s_setpc_b64 s[30:31]` for its return (nothing else
    # in these libraries computes a branch).  Which callee a call reaches is a register value, so the walk is conservative: in front of
    # the instruction after ANY call sit the tails of ALL called functions' returns (and, as if the callee were empty, the call site's own
    # predecessors); in front of a called function's first instruction sit ALL call sites.
    # Long branches: a kernel of more than 128 KB of code (the soft-state-bound instantiations with two obstacle chunks) gets jumps beyond the
    # 16-bit offset of s_branch relaxed into  s_getpc_b64 s[a:b] / s_add_u32 sa, sa, LO / s_addc_u32 sb, sb, HI / s_setpc_b64 s[a:b]:
    # an unconditional branch to (address of the instruction after s_getpc) + (HI:LO).  Recognised as exactly that sequence, nothing looser.
    long_branch = set()
    for i, (a, mn, ops) in enumerate(prog):
        if mn != "s_setpc_b64" or i < 3:
            continue
        g, lo, hi = prog[i - 3], prog[i - 2], prog[i - 1]
        m = re.match(r"s\[(\d+):(\d+)\]$", ops.strip())
        if not (m and g[1] == "s_getpc_b64" and g[2].strip() == ops.strip() and lo[1] == "s_add_u32" and hi[1] == "s_addc_u32"):
            continue
        ra, rb = m.group(1), m.group(2)
        ml = re.match(r"s%s, s%s, (0x[0-9a-f]+|\d+)$" % (ra, ra), lo[2].strip())
        mh = re.match(r"s%s, s%s, (0x[0-9a-f]+|\d+|-1)$" % (rb, rb), hi[2].strip())
        if not (ml and mh):
            continue
        off = int(ml.group(1), 0) + ((int(mh.group(1), 0) & 0xffffffff) << 32)
        if off >= 1 << 63:
            off -= 1 << 64
        tgt = addr_index.get(lo[0] + off)
        if tgt is None:
            continue
        branches_to.setdefault(tgt, []).append(i)
        long_branch.add(i)
    calls = [i for i, (_, mn, _) in enumerate(prog) if mn == "s_swappc_b64"]
    returns = [i for i, (_, mn, _) in enumerate(prog) if mn == "s_setpc_b64" and i not in long_branch]
    kernels = {funcs[i] for i, (_, mn, _) in enumerate(prog) if mn == "s_endpgm"}
    entry_of = {}       # index of a called function's first instruction -> True
    for i in range(len(prog)):
        if (i == 0 or funcs[i - 1] != funcs[i]) and funcs[i] not in kernels:
            entry_of[i] = True
    for r in returns:   # a return must be the plain `s_setpc_b64 s[30:31]` of a called function, not a computed jump inside a kernel
        if funcs[r] in kernels or "s[30:31]" not in prog[r][2]:
            unknown_branch = True
    is_target = set(branches_to)

    def writers_before(i, need, depth=0):
        """yield (index, mnemonic, ops) of the instructions occupying the `need` wait states before instruction i, on every path"""
        if need <= 0 or depth > 8:
            return
        # paths arriving by branch at i
        if i in is_target:
            for src in branches_to[i]:
                # the branch itself takes a wait state
                yield from writers_before(src, need - 1, depth + 1)
                yield (src, prog[src][1], prog[src][2])
        j = i - 1
        if j < 0 or funcs[j] != funcs[i]:
            if i in entry_of:   # first instruction of a called function: every call site leads here
                for c in calls:
                    yield (c, prog[c][1], prog[c][2])
                    yield from writers_before(c, need - 1, depth + 1)
            return
        mn, ops = prog[j][1], prog[j][2]
        if mn in ("s_branch", "s_endpgm", "s_setpc_b64"):
            return      # no fall-through
        yield (j, mn, ops)
        if mn == "s_swappc_b64":   # the call has returned: whatever the callee did last came in between
            for r in returns:
                yield (r, prog[r][1], prog[r][2])
                yield from writers_before(r, need - 2, depth + 1)
        yield from writers_before(j, need - wait_states(mn, ops), depth)

    bad = []
    count = 0
    for i, (a, mn, ops) in enumerate(prog):
        if mn not in DPP_ASM:
            continue
        count += 1
        toks = [t.strip() for t in ops.split(",")]
        src0 = regs(toks[1])
        for (j, wmn, wops) in writers_before(i, WAIT_STATES):
            if not is_valu(wmn):
                continue
            wt = [t.strip() for t in wops.split(",")]
            if not wt:
                continue
            dst = regs(wt[0])
            if wmn in TWO_DEST and len(wt) > 1:
                dst |= regs(wt[1])
            if wmn.startswith("v_cmp") or wmn.startswith("v_readlane") or wmn.startswith("v_readfirstlane"):
                dst = set()
            if dst & src0:
                bad.append((funcs[i], a, ops, prog[j][0], wmn, wops))
        for (j, wmn, wops) in writers_before(i, EXEC_WAIT_STATES):
            if writes_exec(wmn, wops):
                bad.append((funcs[i], a, ops, prog[j][0], wmn, wops + "   (VALU write of EXEC within 5 wait states)"))
    return count, bad, unknown_branch


def check_library(path, expect_fused=True):
    """(number of hand-placed DPP instructions, list of violation strings) for one library / code object.
    expect_fused: the library holds the hand-placed v_fmac_f64_dpp of lanes::fma_bc*, so finding NO such instruction means the
    disassembly was not understood (objdump format change) - a violation, not a pass."""
    cos, tmp = code_objects(path)
    total, msgs = 0, []
    try:
        if not cos:
            msgs.append("no gfx950 code object found")
        for co in cos:
            n, bad, unk = check(co)
            total += n
            for (fn, a, ops, wa, wmn, wops) in bad:
                msgs.append(f"{fn}: {a:#x} v_fmac_f64_dpp {ops}   <-   {wa:#x} {wmn} {wops}")
            if unk and n:
                msgs.append("a branch the checker cannot follow next to hand-placed DPP code")
        if cos and expect_fused and total == 0:
            msgs.append("no v_fmac_f64_dpp found in a fused build: the disassembly was not parsed, nothing was checked")
    finally:
        if tmp:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    return total, msgs


def main(paths):
    rc = 0
    for p in paths:
        cos, tmp = code_objects(p)
        if not cos:
            print(f"{p}: no gfx950 code object found")
            rc = 1
        total = 0
        for co in cos:
            n, bad, unk = check(co)
            total += n
            for (fn, a, ops, wa, wmn, wops) in bad:
                print(f"{p}: HAZARD in {fn}: {a:#x} v_fmac_f64_dpp {ops}   <-   {wa:#x} {wmn} {wops}")
                rc = 1
            if unk and n:
                print(f"{p}: a branch the checker cannot follow next to hand-placed DPP code")
                rc = 1
        print(f"{p}: {total} hand-placed DPP instruction(s) checked, {'violations found' if rc else 'no hazard'}")
        if tmp:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
