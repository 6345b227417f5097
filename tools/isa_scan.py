#!/usr/bin/env python3
"""Static checks on the gfx950 code object inside liblamslide_hip.so (no GPU needed).

Why: two of this library's defects were ordering hazards that no functional test sees reliably (profiles/r02_resident.txt: an inline-asm
VALU instruction reading an MFMA result without wait states; run-to-run differences when weight-refill loads sat 2-3 instructions behind
the MFMAs whose registers they overwrite), and the token-stationary linear1 kernel (csrc/k_lin1.hip.h) issues its LDS-DMA and its stores
from inline asm, where hipcc pads nothing.  The rules below are checked over the disassembly of every kernel:

  R1  no vector-memory register load (global_load / buffer_load / scratch_load / flat_load) writes a register that was the C/D operand of
      an MFMA issued fewer than `mfma_load_gap` (12) instructions earlier in the same branch-free run of instructions.  LDS reads in that position are counted
      but not failed: hipcc places ds_read directly behind an MFMA that read the same registers as its C operand in every MFMA kernel
      of this library (its hazard recognizer pads the wait states that pattern needs); the unexplained k_resident nondeterminism came and
      went with the distance of GLOBAL loads.
  R2  a vector-memory instruction that takes a scalar base (global_load_lds / global_store / global_load with an s[..] operand) has at
      least 5 wait states between the last VECTOR-instruction write of that base (v_readfirstlane, v_readlane, v_cmp) and itself
      ("VALU writes SGPR -> VMEM reads it"; scalar-ALU writes are interlocked).
  R3  an LDS-DMA instruction (global_load_lds_*, buffer_load ... lds) has at least 1 wait state after the last write of m0.
  R4  (report only) kernels that use scratch.

Usage: tools/isa_scan.py [path/to/liblamslide_hip.so]   -> prints a per-kernel table, exit status 1 on a violation."""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile
from dataclasses import dataclass, field

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(so_path: str) -> str:
    """llvm-objdump --offloading writes the bundles next to its input: work on a copy in a scratch directory."""
    with tempfile.TemporaryDirectory() as td:
        dst = os.path.join(td, "lib.so")
        shutil.copy(so_path, dst)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], check=True, capture_output=True, cwd=td)
        objs = [f for f in os.listdir(td) if "gfx950" in f]
        if not objs:
            raise RuntimeError("no gfx950 code object in " + so_path)
        res = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", os.path.join(td, objs[0])], check=True,
                             capture_output=True, text=True)
        return res.stdout


_REG = re.compile(r"\b([vsa])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(operand: str):
    """register set named by one operand: {('v', 12), ...}"""
    out = set()
    for m in _REG.finditer(operand):
        kind = m.group(1)
        if m.group(4) is not None:
            out.add((kind, int(m.group(4))))
        else:
            out.update((kind, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    if re.search(r"\bm0\b", operand):
        out.add(("m0", 0))
    if re.search(r"\bvcc\b", operand):
        out.update({("s", 106), ("s", 107)})
    return out


@dataclass
class Inst:
    op: str
    operands: list
    text: str


@dataclass
class Report:
    name: str
    n_inst: int = 0
    n_mfma: int = 0
    r1: list = field(default_factory=list)
    r1_lds: int = 0
    r2: list = field(default_factory=list)
    r3: list = field(default_factory=list)
    scratch: int = 0
    min_mfma_load_gap: int = 10 ** 9


def parse(dis: str):
    kernels, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append(Inst(parts[0], ops, body))
    return kernels


def is_reg_load(i: Inst) -> bool:
    if i.op.startswith("ds_read") or i.op.startswith("ds_load"):
        return True
    if re.match(r"(global|flat|scratch|buffer)_load", i.op):
        return "_lds_" not in i.op and not any(o.split()[-1:] == ["lds"] or o == "lds" for o in i.operands)
    return False


def is_lds_dma(i: Inst) -> bool:
    return "_load_lds_" in i.op or (i.op.startswith("buffer_load") and any("lds" in o.split() for o in i.operands))


def wait_states(i: Inst) -> int:
    if i.op == "s_nop":
        return int(i.operands[0], 0) + 1
    return 1


def scan(name: str, insts, mfma_load_gap: int) -> Report:
    r = Report(name, n_inst=len(insts))
    recent_mfma = []  # (index, C/D register set)
    last_sgpr_write = {}  # sgpr index -> wait-state clock of its last SALU / readfirstlane write
    last_m0_write = None
    clock = 0
    for idx, i in enumerate(insts):
        op = i.op
        if op.startswith("scratch_"):
            r.scratch += 1
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            # the listing is in layout order, not execution order: behind a branch the next instruction belongs to another path (hipcc's
            # structurizer puts never-fallen-through "Flow" blocks between a loop body and its successor).  R1 is a straight-line rule.
            recent_mfma = []
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            r.n_mfma += 1
            cd = regs(i.operands[0]) | (regs(i.operands[3]) if len(i.operands) > 3 else set())
            recent_mfma.append((idx, cd))
            recent_mfma = recent_mfma[-16:]
        elif is_reg_load(i):
            dst = regs(i.operands[0])
            for midx, cd in recent_mfma:
                gap = idx - midx
                if dst & cd:
                    r.min_mfma_load_gap = min(r.min_mfma_load_gap, gap)
                    if gap < mfma_load_gap:
                        if i.op.startswith("ds_"):
                            r.r1_lds += 1
                        else:
                            r.r1.append(f"{i.text}   <- {gap} instructions behind {insts[midx].text}")
        # R2 / R3: consumers first (a write by this very instruction does not count against it)
        if re.match(r"(global|buffer)_(load|store|atomic)", op):
            sbase = set()
            for o in i.operands:
                if re.fullmatch(r"s\[\d+:\d+\]", o.split()[0]):
                    sbase |= {n for k, n in regs(o.split()[0]) if k == "s"}
            for n in sbase:
                if n in last_sgpr_write and clock - last_sgpr_write[n] < 5:
                    r.r2.append(f"{i.text}   <- s{n} written {clock - last_sgpr_write[n]} wait states earlier")
                    break
            if is_lds_dma(i) and last_m0_write is not None and clock - last_m0_write < 1:
                r.r3.append(f"{i.text}   <- m0 written {clock - last_m0_write} wait states earlier")
        # producers: SGPRs written by a VECTOR instruction (v_readfirstlane / v_readlane / v_cmp into an SGPR pair).  Scalar-ALU writes are
        # interlocked on gfx9 (hipcc itself emits s_add_u32 / s_addc_u32 directly in front of the global_load that takes the pair as its base)
        if op in ("v_readfirstlane_b32", "v_readlane_b32") or (op.startswith("v_cmp") and i.operands and i.operands[0].startswith("s[")):
            for k, n in regs(i.operands[0]):
                if k == "s":
                    last_sgpr_write[n] = clock + wait_states(i)
        if i.operands and re.search(r"\bm0\b", i.operands[0]) and op.startswith("s_") and not op.startswith(("s_cmp", "s_bitcmp")):
            last_m0_write = clock + wait_states(i)
        clock += wait_states(i)
    return r


def scan_library(so_path: str, mfma_load_gap: int = 12):
    kernels = parse(disassemble(so_path))
    return [scan(n, insts, mfma_load_gap) for n, insts in kernels.items()]


def demangle_short(name: str) -> str:
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip()
        return re.sub(r"\(.*$", "", out)[:100]
    except Exception:
        return name[:100]


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lam_slide_amd",
                                                            "liblamslide_hip.so")
    bad = 0
    for r in scan_library(so):
        if r.n_mfma == 0 and not (r.r2 or r.r3):
            continue
        gap = "-" if r.min_mfma_load_gap > 10 ** 8 else str(r.min_mfma_load_gap)
        print(f"{demangle_short(r.name):100s} inst {r.n_inst:6d} mfma {r.n_mfma:5d} min load-behind-MFMA gap {gap:>3s} scratch ops {r.scratch:3d} "
              f"R1 {len(r.r1)} (lds {r.r1_lds}) R2 {len(r.r2)} R3 {len(r.r3)}")
        for v in (r.r1 + r.r2 + r.r3)[:6]:
            print("      " + v)
        bad += len(r.r1) + len(r.r2) + len(r.r3)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
