#!/usr/bin/env python3
"""Mechanical checks of the gfx950 code objects this package ships: the inline-asm GEMMs depend on
compiler behaviour that a hipcc bump can change silently, so the resource usage and three ISA patterns are checked on the
BUILT code (no GPU needed; hipcc cross-compiles, llvm-readelf / llvm-objdump read the result).

  kernel_metadata(co)   per kernel: VGPRs, AGPRs, SGPRs, scratch bytes, spills, LDS bytes   (llvm-readelf --notes)
  disassemble(co)       per kernel: the instruction list                                      (llvm-objdump -d)
  find_flat(...)        flat_load / flat_store where a global_ access was written (a base pointer that lost its address space)
  find_sgpr_hazards()   a VALU-written SGPR read as the scalar base of a VMEM access within 5 wait states (the hazard hipcc
                        does not pad for an inline-asm consumer: the first split-dW build faulted on it)
  find_inflight_touch() any instruction that reads or writes a VGPR while a ds_read into it is still outstanding (no
                        covering s_waitcnt lgkmcnt yet): hipcc's "re-pack" of bf16x8 fragments (v_lshrrev / v_perm pairs) between
                        an asm LDS read and its wait mixed stale and new halves (round 2: dW 3 % off under concurrent load)

usage: python tools/isa_guard.py [object ...]    (default: every object under npi_gnn_amd/build) -> a report, exit 1 on findings
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def _tool(name: str) -> str:
    p = os.path.join(LLVM, name)
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


def code_object(obj: str, out_dir: str) -> str:
    """the gfx950 code object embedded in a host object file built by hipcc (.hip_fatbin -> unbundle)"""
    base = os.path.join(out_dir, os.path.basename(obj))
    subprocess.check_call([_tool("llvm-objcopy"), f"--dump-section=.hip_fatbin={base}.fatbin", obj, base + ".stripped"])
    subprocess.check_call([_tool("clang-offload-bundler"), "--unbundle", "--type=o", f"--targets={TARGET}",
                           f"--input={base}.fatbin", f"--output={base}.co"])
    return base + ".co"


def demangle(names):
    import shutil
    filt = shutil.which("c++filt")
    if not filt:
        return list(names)
    p = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True)
    return p.stdout.splitlines()


def kernel_metadata(co: str) -> dict:
    """{mangled kernel name: {vgpr, agpr, sgpr, scratch, lds, vgpr_spill, sgpr_spill, max_wg}}"""
    import yaml
    txt = subprocess.run([_tool("llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    start, end = txt.index("---"), txt.rindex("...")
    doc = yaml.safe_load(txt[start:end])
    out = {}
    for k in doc["amdhsa.kernels"]:
        out[k[".name"]] = {"vgpr": k[".vgpr_count"], "agpr": k.get(".agpr_count", 0), "sgpr": k[".sgpr_count"],
                           "scratch": k[".private_segment_fixed_size"], "lds": k[".group_segment_fixed_size"],
                           "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                           "dynamic_stack": bool(k.get(".uses_dynamic_stack", False)),
                           "max_wg": k.get(".max_flat_workgroup_size")}
    return out


_FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:$")
_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")


class Code(list):
    """instruction texts of one kernel; ``addr[i]`` = byte address of instruction i"""

    def __init__(self):
        super().__init__()
        self.addr = []


def disassemble(co: str) -> dict:
    """{symbol: Code} -- instruction text without the encoding comment, plus every instruction's address"""
    txt = subprocess.run([_tool("llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = out.setdefault(m.group(1), Code())
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if ins:
            a = _ADDR.search(line)
            cur.append(ins)
            cur.addr.append(int(a.group(1), 16) if a else -1)
    return out


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _regs(rx, text):
    s = set()
    for m in rx.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def _split(ins):
    op, _, rest = ins.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    return op, ops


_VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load",
         "scratch_store", "flat_load", "flat_store", "flat_atomic")
_BRANCH = ("s_branch", "s_cbranch")
_END = ("s_endpgm", "s_setpc", "s_swappc", "s_trap")


def control_flow(code: Code):
    """(block leaders sorted, successors {leader index: [leader index, ...]}) from the branch offsets (simm16 dwords
    relative to the next instruction)"""
    index_of = {a: i for i, a in enumerate(code.addr)}
    target = {}
    for i, ins in enumerate(code):
        op, ops = _split(ins)
        if op.startswith(_BRANCH) and ops:
            off = int(ops[0], 0)
            if off >= 0x8000:
                off -= 0x10000
            t = index_of.get(code.addr[i] + 4 + 4 * off)
            if t is not None:
                target[i] = t
    leaders = {0} | set(target.values()) | {i + 1 for i, ins in enumerate(code)
                                            if ins.startswith(_BRANCH + _END) and i + 1 < len(code)}
    leaders = sorted(leaders)
    succ = {}
    for k, lead in enumerate(leaders):
        end = (leaders[k + 1] if k + 1 < len(leaders) else len(code)) - 1
        last = code[end]
        nxt = []
        if last.startswith("s_branch"):
            nxt = [target[end]] if end in target else []
        elif last.startswith("s_cbranch"):
            nxt = ([target[end]] if end in target else []) + ([end + 1] if end + 1 < len(code) else [])
        elif last.startswith(_END):
            nxt = []
        elif end + 1 < len(code):
            nxt = [end + 1]
        succ[lead] = nxt
    return leaders, succ


def find_flat(instrs):
    return [(i, ins) for i, ins in enumerate(instrs) if ins.startswith(("flat_load", "flat_store", "flat_atomic"))]


def find_sgpr_hazards(instrs, wait_states: int = 5):
    """VALU writes SGPR -> VMEM reads that SGPR needs `wait_states` wait states in between (CDNA ISA, data hazards).
    Straight-line windows only (a taken branch costs more than the window)."""
    found = []
    for i, ins in enumerate(instrs):
        op, ops = _split(ins)
        if not op.startswith("v_") or not ops:
            continue
        if ops[0].startswith("v"):
            continue
        written = _regs(_SREG, ops[0])                         # first operand = destination (v_readfirstlane, v_cmp_e64, ...)
        if not written:
            continue
        ws = 0
        for j in range(i + 1, len(instrs)):
            nop, nops = _split(instrs[j])
            if ws >= wait_states:
                break
            if nop.startswith(_VMEM) and written & _regs(_SREG, " ".join(nops)):
                found.append((i, ins, j, instrs[j], ws))
                break
            if nop.startswith(_BRANCH + _END):
                break
            ws += (int(nops[0], 0) + 1) if nop == "s_nop" and nops else 1
    return found


_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
_CAP = 15
_LDS_DEST = ("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append", "ds_ordered_count")
_SMEM = ("s_load", "s_buffer_load", "s_scratch_load", "s_memtime", "s_memrealtime", "s_atc_probe", "s_dcache")


def find_inflight_touch(code: Code):
    """A VGPR that is the destination of a ds_read still in flight must not be read or written by anything else.

    Forward may-analysis over the kernel's control-flow graph.  State: for every VGPR that may hold an outstanding LDS
    return, the least number of LGKM operations issued after it on any path (LDS operations return in order, so
    `s_waitcnt lgkmcnt(n)` retires every entry with at least n younger operations -- unless a scalar memory load, which
    returns out of order, may be outstanding too: then only lgkmcnt(0) retires anything).  Join = union, min, or."""
    if not len(code):
        return []
    leaders, succ = control_flow(code)
    bounds = {lead: (leaders[k + 1] if k + 1 < len(leaders) else len(code)) for k, lead in enumerate(leaders)}
    entry = {lead: None for lead in leaders}                    # None = not reached yet
    entry[0] = ({}, False, {})                                  # (reg -> younger count, smem outstanding, reg -> issuing index)
    found = {}

    def step(i, state, report):
        regs, smem, src = state
        ins = code[i]
        op, ops = _split(ins)
        if op == "s_waitcnt":
            m = _LGKM.search(ins)
            if m:
                n = int(m.group(1))
            elif len(ops) == 1 and re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", ops[0]):
                n = (int(ops[0], 0) >> 8) & 0xF                # gfx9 encoding: lgkmcnt in bits 11:8
            else:
                return state
            if smem:
                if n == 0:
                    return {}, False, {}
                return state
            keep = {r: c for r, c in regs.items() if c < n}
            return keep, False, {r: src[r] for r in keep}
        is_lds = op.startswith("ds_")
        is_smem = op.startswith(_SMEM)
        if regs:
            # an LDS instruction may name an in-flight register as its DESTINATION (returns are in order); everything else may not
            has_dest = is_lds and (op.startswith(_LDS_DEST) or "_rtn" in op)
            used = _regs(_VREG, " ".join(ops[1:] if has_dest else ops))
            hit = used & set(regs)
            # The matrix instructions consume fragments behind COUNTED waits whose sufficiency depends on which of the
            # conditional prefetches ran -- correlated branches a path-insensitive analysis cannot follow (the numerics
            # tests own that).  Everything else -- VALU re-packs (v_perm / v_lshrrev), moves, stores -- has no business
            # with a register whose LDS return is outstanding on ANY path.
            if hit and report and not op.startswith(("v_mfma", "v_smfmac")):
                r = min(hit)
                found.setdefault(i, (i, ins, src[r], code[src[r]], sorted(hit)))
        if is_lds or is_smem:
            regs = {r: min(c + 1, _CAP) for r, c in regs.items()}
            src = dict(src)
            if is_lds and ops and (op.startswith(_LDS_DEST) or "_rtn" in op):
                for r in _regs(_VREG, ops[0]):
                    regs[r] = 0
                    src[r] = i
            return regs, smem or is_smem, src
        return state

    def join(a, b):
        if a is None:
            return b, True
        regs, smem, src = dict(a[0]), a[1] or b[1], dict(a[2])
        changed = smem != a[1]
        for r, c in b[0].items():
            if r not in regs or c < regs[r]:
                regs[r], src[r] = c, b[2][r]
                changed = True
        return (regs, smem, src), changed

    work = [0]
    while work:
        lead = work.pop()
        state = entry[lead]
        for i in range(lead, bounds[lead]):
            state = step(i, state, False)
        for nxt in succ[lead]:
            merged, changed = join(entry[nxt], state)
            if changed:
                entry[nxt] = merged
                work.append(nxt)
    for lead in leaders:                                        # second pass over the fixed point: report
        state = entry[lead]
        if state is None:
            continue
        for i in range(lead, bounds[lead]):
            state = step(i, state, True)
    return [found[i] for i in sorted(found)]


def find_inflight_touch_linear(code: Code, limit: int = 600):
    """The same rule along the FALL-THROUGH path only: from every ds_read forward -- conditional forward branches not
    taken -- until the `s_waitcnt lgkmcnt` that retires it by count, an unconditional branch, the backward branch that
    closes its loop, or `limit` instructions.  Exact on the path it walks (no correlated-branch false positives), so the
    matrix instructions are checked here as well.  This is the pattern of the round-2 bug: a VALU re-pack in the straight
    line between an asm LDS read and its wait."""
    found = {}
    addr = code.addr
    for i, ins in enumerate(code):
        op, ops = _split(ins)
        if not (op.startswith(_LDS_DEST) and ops):
            continue
        dest = _regs(_VREG, ops[0])
        younger, smem = 0, False
        for j in range(i + 1, min(i + 1 + limit, len(code))):
            nop, nops = _split(code[j])
            if nop == "s_waitcnt":
                m = _LGKM.search(code[j])
                if m:
                    n = int(m.group(1))
                    if (not smem and younger >= n) or n == 0:
                        break
                continue
            if nop.startswith("s_branch") or nop.startswith(_END):
                break
            if nop.startswith("s_cbranch") and nops:
                off = int(nops[0], 0)
                if off >= 0x8000:
                    break                                       # backward: the end of this read's loop body
                continue
            is_lds = nop.startswith("ds_")
            has_dest = is_lds and (nop.startswith(_LDS_DEST) or "_rtn" in nop)
            hit = _regs(_VREG, " ".join(nops[1:] if has_dest else nops)) & dest
            if hit:
                found.setdefault(j, (j, code[j], i, ins, sorted(hit)))
            if is_lds:
                younger += 1
            elif nop.startswith(_SMEM):
                smem = True
    return [found[j] for j in sorted(found)]


def analyse(obj: str, tmp: str):
    co = code_object(obj, tmp)
    meta = kernel_metadata(co)
    code = disassemble(co)
    return meta, code


STRICT = False


def main(argv):
    global STRICT
    if "--strict" in argv:
        STRICT = True
        argv = [a for a in argv if a != "--strict"]
    objs = argv or sorted(os.path.join(ROOT, "npi_gnn_amd", "build", f) for f in os.listdir(os.path.join(ROOT, "npi_gnn_amd", "build"))
                          if f.endswith(".o"))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            meta, code = analyse(obj, tmp)
            names = list(meta)
            pretty = dict(zip(names, demangle(names)))
            print(f"== {os.path.basename(obj)}: {len(meta)} kernels")
            for name, m in sorted(meta.items(), key=lambda kv: -kv[1]["vgpr"])[:12]:
                print(f"   vgpr {m['vgpr']:3d} agpr {m['agpr']:3d} sgpr {m['sgpr']:3d} lds {m['lds']:6d} scratch {m['scratch']:4d}  {pretty[name][:110]}")
            for name, m in meta.items():
                if m["scratch"] or m["vgpr_spill"] or m["dynamic_stack"]:
                    print(f"   SCRATCH/SPILL {m}  {pretty[name]}")
                    bad += 1
                elif m["sgpr_spill"]:                           # SGPRs parked in VGPR lanes: no memory traffic, listed only
                    print(f"   (sgpr spills to VGPR lanes: {m['sgpr_spill']})  {pretty[name][:100]}")
            for sym, instrs in code.items():
                for what, hits in (("flat access", find_flat(instrs)), ("VALU->SGPR->VMEM hazard", find_sgpr_hazards(instrs)),
                                   ("touch of an in-flight ds_read destination (fall-through path)", find_inflight_touch_linear(instrs)),
                                   ("touch of an in-flight ds_read destination (any CFG path, --strict)",
                                    find_inflight_touch(instrs) if STRICT else [])):
                    if what == "flat access" and not any(k in sym for k in ("gemm_split_ws", "gemm_dw_split", "gemm_bf16_ws")):
                        continue
                    for h in hits[:5]:
                        print(f"   {what} in {sym}: {h}")
                    bad += len(hits)
    print("findings:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
