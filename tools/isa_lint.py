#!/usr/bin/env python3
"""tools/isa_lint.py — a hazard lint on the MACHINE CODE of the library (every gfx950 code object of a fat binary, or a device-only
object), for the data hazards the compiler's hazard recognizer cannot apply inside an inline-asm string (it treats INLINEASM as opaque)
and which therefore depend on the code layout around every `asm volatile` memory instruction of the kernels:

  A  a VMEM instruction (buffer_ / global_ / flat_ / scratch_ load, store or atomic) reads an SGPR that a VALU instruction
     (v_readfirstlane_b32, v_readlane_b32, a VOP3 compare or a carry-out with an SGPR destination ...) wrote fewer than FIVE wait states
     before it                                         (gfx9 / gfx940 ISA guide, "VALU writes SGPR -> VMEM reads that SGPR: 5 wait states")
  B  a VMEM / FLAT store of more than 64 bits of data is followed IMMEDIATELY by a VALU instruction that writes one of its data VGPRs
                                                       ("VMEM store more than 64 bits followed by a write of the writedata VGPRs: 1 wait state")

Wait states: every instruction counts one, `s_nop N` counts N + 1.  The scan is linear over a function's instructions; a label (branch
target) keeps the window of its fall-through predecessor — the straight-line approximation the findings of round 3 were all within.
Used by tests/test_isa_guard.py; `python tools/isa_lint.py <file>` prints the findings of one library / object."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VMEM = re.compile(r"^(buffer|global|flat|scratch|tbuffer)_(load|store|atomic)")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def code_objects(path):
    """gfx950 code objects of a fat binary (one __CLANG_OFFLOAD_BUNDLE__ per translation unit); a plain ELF is returned as it is"""
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        o = m.start()
        n, = struct.unpack_from("<Q", blob, o + 24)
        p = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                out.append(blob[o + off:o + off + size])
    return out or [blob]


def disassemble(path):
    texts = []
    with tempfile.TemporaryDirectory() as d:
        for i, data in enumerate(code_objects(path)):
            f = os.path.join(d, "co%d.o" % i)
            open(f, "wb").write(data)
            r = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-400:])
            texts.append(r.stdout)
    return texts


def _regs(rx, text):
    out = set()
    for m in rx.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _operands(line):
    """mnemonic and operand strings of a disassembly line ('\tv_readfirstlane_b32 s4, v1   // 000...')"""
    body = line.split("//")[0].strip()
    if not body:
        return None, []
    parts = body.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def valu_sgpr_dests(mn, ops):
    """SGPRs a VALU instruction writes: its first operand when that is scalar (v_readfirstlane / v_readlane / VOP3 compares),
    and the second one of the carry-out / scale forms (v_add_co_u32 v1, s[2:3], ... ; v_div_scale_f32 v1, s[2:3], ...)"""
    if not mn.startswith("v_") or not ops:
        return set()
    d = _regs(SREG, ops[0]) if ops[0].startswith("s") else set()
    if len(ops) > 1 and ops[1].startswith("s") and (re.search(r"_co_|v_div_scale|v_mad_[ui]64", mn)):
        d |= _regs(SREG, ops[1])
    return d


def store_data_vgprs(mn, ops):
    """data VGPRs of a store wider than 64 bits (dwordx3 / dwordx4 / b96 / b128), else empty"""
    if not re.match(r"^(buffer|global|flat|scratch)_store_(dwordx[34]|b96|b128)", mn) or not ops:
        return set()
    # global_store: vaddr, vdata, saddr | buffer_store: vdata, vaddr, srsrc, soffset | flat_store: vaddr, vdata | scratch_store: vaddr, vdata, saddr
    data = ops[0] if mn.startswith("buffer") else (ops[1] if len(ops) > 1 else "")
    return _regs(VREG, data)


def lint(text):
    """findings of one code object's disassembly: list of (function, rule, instruction, detail)"""
    out = []
    func = "?"
    window = []          # (wait states this instruction takes, SGPRs it VALU-wrote, its text)
    pending_store = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            if not name.startswith("L"):       # a function symbol, not a local label
                func, window, pending_store = name, [], None
            continue
        mn, ops = _operands(line)
        if not mn or mn.startswith("."):
            continue
        # ---- rule B: the instruction right after a wide store
        if pending_store is not None:
            data, stext = pending_store
            if mn.startswith("v_") and ops:
                dst = _regs(VREG, ops[0])
                if dst & data:
                    out.append((func, "B", stext, "next instruction %s writes its data register(s) %s" % (" ".join([mn] + ops[:1]), sorted(dst & data))))
            pending_store = None
        # ---- rule A: VMEM reading an SGPR a VALU wrote < 5 wait states ago
        if VMEM.match(mn):
            sread = set()
            for o in ops:
                sread |= _regs(SREG, o)
            ws = 0
            for (w, sw, t) in reversed(window):
                if ws >= 5:
                    break
                if sw & sread:
                    out.append((func, "A", " ".join([mn] + ops), "%d wait state(s) after %s" % (ws, t)))
                    break
                ws += w
            data = store_data_vgprs(mn, ops)
            if data:
                pending_store = (data, " ".join([mn] + ops))
        w = 1
        if mn == "s_nop":
            try:
                w = int(ops[0], 0) + 1
            except (ValueError, IndexError):
                w = 1
        window.append((w, valu_sgpr_dests(mn, ops), " ".join([mn] + ops[:2])))
        if len(window) > 8:
            window.pop(0)
    return out


def lint_file(path):
    res = []
    for t in disassemble(path):
        res += lint(t)
    return res


if __name__ == "__main__":
    bad = lint_file(sys.argv[1])
    for f, rule, ins, why in bad:
        print("%s  rule %s  %s  — %s" % (f[:70], rule, ins, why))
    print("%d finding(s)" % len(bad))
    sys.exit(1 if bad else 0)
