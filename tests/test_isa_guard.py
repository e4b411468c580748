"""Guards on the machine code of the library that ships to the GPU box (gmat_amd/lib/libgmat_hip.so), checked here without
a GPU: the gfx950 code objects are cut out of the fat binary and disassembled with ROCm's llvm-objdump.

v_ashr_pk_u8_i32 / v_ashr_pk_i8_i32: hipcc (ROCm 7.2) selects them for clip(v >> s) of two neighbouring values and then
treats the upper 16 bits of the result as zero; on gfx950 the instruction leaves them as they were.  The CPU emulation of
the kernels (plain C++ semantics) cannot see that, the bytes only go wrong on hardware — twice so far: packed RGB in round 1
(px_math.h luma_chan) and every output dword of scale_yuv2p_kernel in round 2, with rgb2yuv444_kernel right only because of
what the register allocator happened to leave in the destination.  The kernels use clip_u8_shr() (clamp, then shift), for
which the compiler does not form the instruction; this test makes sure it stays that way for every kernel, present and future."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gmat_amd", "lib", "libgmat_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FORBIDDEN = ("v_ashr_pk_u8_i32", "v_ashr_pk_i8_i32")


def gfx950_code_objects(path):
    """the entries of every __CLANG_OFFLOAD_BUNDLE__ in the file whose target is gfx950 (one bundle per .hip translation unit)"""
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
    return out


@pytest.fixture(scope="module")
def disassembly():
    if not os.path.exists(OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    assert os.path.exists(LIB), "build first (__graft_entry__.build())"
    objs = gfx950_code_objects(LIB)
    texts = []
    with tempfile.TemporaryDirectory() as d:
        for i, data in enumerate(objs):
            f = os.path.join(d, f"co{i}.o")
            open(f, "wb").write(data)
            r = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-400:]
            texts.append(r.stdout)
    return texts


def test_the_extraction_sees_the_kernels(disassembly):
    """not vacuous: one code object per kernel source, and the kernels the product launches are in them"""
    import glob
    nsrc = len(glob.glob(os.path.join(ROOT, "gmat_amd", "csrc", "k_*.hip")))
    assert len(disassembly) >= nsrc, (len(disassembly), nsrc)
    allt = "\n".join(disassembly)
    for k in ("scale_yuv2s_kernel", "scale_yuv2s_np_kernel", "scale_yuv2p_kernel", "scale_yuv2px_kernel", "scale_yuv1x2_kernel", "scale_yuv3x1_kernel", "scale_yuv3r_kernel", "scale_yuv3x2_kernel", "scale_yuv32r_kernel", "scale_yuv4r_kernel", "scale_yuv4x1_kernel", "scale_rgb2s_kernel", "scale_rgb2h_kernel", "scale_rgb2y_kernel", "rgb2yuv420s_kernel", "rgb2yuv444_kernel", "smooth121_kernel",
              "scale_yuv_kernel", "yuv2rgb_kernel"):
        assert k in allt, k
    assert allt.count("v_dot2") > 1000 and "v_perm_b32" in allt


def test_no_packed_shift_clamp_instructions(disassembly):
    hits = []
    for t in disassembly:
        func = "?"
        for line in t.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                func = m.group(1)
            elif any(f in line for f in FORBIDDEN):
                hits.append(func)
    by = {}
    for f in hits:
        by[f] = by.get(f, 0) + 1
    assert not hits, f"v_ashr_pk_* (upper half of the result is NOT zero on gfx950) in: {by} — use clip_u8_shr()"


def test_no_other_instruction_leaves_part_of_its_destination_untouched(disassembly):
    """Audit of round 2 (all code objects): every SDWA instruction pads the unselected destination bits with zeros
    (dst_unused:UNUSED_PAD), the d16_hi forms present are stores, and there is no v_sat_pk_* / v_cvt_pk_u8_* / d16 load.
    Anything that MERGES into its destination is flagged here so that it gets looked at on hardware before it ships —
    the emulator cannot tell."""
    merging = ("UNUSED_PRESERVE", "v_sat_pk_u8_i16", "v_cvt_pk_u8_f32")
    d16_loads = re.compile(r"\b(global|flat|buffer|ds|scratch)_(load|read)_\w*d16")
    # Verified on the GPU and listed (round 3): v_sat_pk_u8_i16 in the strip kernels' colour stage (s2_sat_pk_u8_i16, inline asm).
    # Its result's UPPER half is never consumed — every use goes straight into a v_perm_b32 that selects bytes 0 and 1 — so the
    # question "zeroed or preserved" does not arise; GPU suite bit-exact with it (profiles/r03c_valu_cuts_ab.txt, 4411 passed).
    verified = {"v_sat_pk_u8_i16": ("scale_yuv2s_kernel", "scale_yuvg_rgb_kernel")}
    hits = {}
    for t in disassembly:
        func = "?"
        for line in t.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                func = m.group(1)
                continue
            if any(m in line for m in merging) or d16_loads.search(line):
                op = line.split()[0]
                ok = [k for k, funcs in verified.items() if op.startswith(k) and any(f in func for f in funcs)]
                if not ok:
                    hits[op + " in " + func[:60]] = hits.get(op + " in " + func[:60], 0) + 1
    assert not hits, f"instructions that keep part of their destination: {hits} — verify on the GPU, then list them here"


def test_strip_kernels_use_no_scratch_memory(disassembly):
    """The register-resident kernels must not touch private (scratch) memory: a runtime index into a register array, pointers
    into the kernel-argument block, or `cond ? P.x : P.y` on members of a struct the compiler keeps in memory each put it there
    silently — scale_yuv1x2_kernel's first build copied its argument block to scratch and ran at 10.4 us per frame.
    (The round-1 tiled kernels scale_rgb_kernel / scale_yuv2x_kernel carry 20 - 188 bytes of it; they are not listed.)"""
    strip = ("scale_yuv2s_kernel", "scale_yuv2s_np_kernel", "scale_yuv2p_kernel", "scale_yuv2px_kernel", "scale_yuv1x2_kernel", "scale_yuv3x1_kernel", "scale_yuv3r_kernel", "scale_yuv3x2_kernel", "scale_yuv32r_kernel", "scale_yuv4r_kernel", "scale_yuv4x1_kernel",
             "scale_rgb2s_kernel", "scale_rgb2h_kernel", "scale_rgb2y_kernel", "rgb2yuv420s_kernel", "smooth121_kernel")
    hits, seen = {}, set()
    for t in disassembly:
        func = None
        for line in t.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                func = m.group(1) if any(k in m.group(1) for k in strip) else None
                if func:
                    seen.add(next(k for k in strip if k in func))
            elif func and re.search(r"\bscratch_(load|store)", line):
                hits[func] = hits.get(func, 0) + 1
    assert seen == set(strip), set(strip) - seen
    assert not hits, f"scratch memory in: {hits}"


def test_streaming_stores_are_where_they_were_measured(disassembly):
    """Non-temporal stores (px_math.h st_stream, profiles/r03zs_nt_stores_ab.txt): in the kernels where a wave writes whole lines — and NOT in
    smooth121_kernel, whose 240-byte columns lose 20 - 30 % with them, nor on any LOAD of the headline (-17 %).  A refactoring that moves a
    store helper from one list to the other changes a measured number silently; this pins the machine code."""
    stores, loads = {}, {}
    for t in disassembly:
        func = None
        for line in t.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                func = m.group(1)
                continue
            if func and re.search(r"\b(global|buffer)_store_\w+", line):
                s = stores.setdefault(func, [0, 0])
                s[1 if re.search(r"\bnt\b", line) else 0] += 1
            if func and re.search(r"\b(global|buffer)_load_\w+", line) and re.search(r"\bnt\b", line):
                loads[func] = loads.get(func, 0) + 1

    def total(key):
        plain = sum(v[0] for f, v in stores.items() if key in f)
        nt = sum(v[1] for f, v in stores.items() if key in f)
        return plain, nt
    for k in ("scale_yuv2s_kernel", "scale_yuv2p_kernel", "scale_yuv3r_kernel", "scale_yuv32r_kernel", "rgb2yuv420s_kernel", "yuv2rgb_kernel",
              "flip_direct_kernel", "median3x3s_kernel"):
        plain, nt = total(k)
        assert nt > 0, (k, plain, nt)
    for k in ("scale_yuvg_", "rotate_lds_kernel"):
        plain, nt = total(k)
        assert plain > 0 and nt == 0, (k, plain, nt)
    # smooth121_kernel<BPP, TRANSPOSED, ...>: the plain smooth (TRANSPOSED = false) never streams; the transposed forms do behind a flag,
    # for destinations whose rows start on 128-byte lines only (smooth121_line_dst), and keep their plain stores for the rest
    for f, v in stores.items():
        m = re.search(r"smooth121_kernelILi\dELb([01])E", f)
        if m and m.group(1) == "0":
            assert v[1] == 0 and v[0] > 0, (f, v)
        elif m:
            assert v[0] > 0, (f, v)
    assert any(re.search(r"smooth121_kernelILi\dELb1E", f) and v[1] > 0 for f, v in stores.items())
    assert not any("scale_yuv2s" in f for f in loads), [f for f in loads if "scale_yuv2s" in f][:2]
    assert any("flip_direct_kernel" in f for f in loads)


def test_inline_asm_memory_instructions_keep_their_distance_from_valu_written_scalars():
    """A memory instruction may read a scalar register a VALU instruction (v_readfirstlane_b32) wrote five wait states later at the earliest; the
    compiler keeps that distance between its own instructions and does not look into an inline-asm string.  The band walker's store had none:
    dormant in the shipped code layout, a GPU fault (no message) as soon as two lines of its kernel's prologue changed (round 3, FINDINGS.md
    R3-walker-bands).  Every inline-asm memory instruction with a scalar operand starts with `s_nop 4`."""
    import glob
    bad = []
    for path in glob.glob(os.path.join(ROOT, "gmat_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "gmat_amd", "csrc", "*.h")):
        for m in re.finditer(r'asm\s+volatile\s*\(\s*"((?:[^"\\]|\\.)*)"([^;]*);', open(path).read()):
            text, operands = m.group(1), m.group(2)
            if re.search(r"\b(global|buffer|flat)_(load|store)", text) and '"s"(' in operands and not text.startswith("s_nop 4"):
                bad.append((os.path.basename(path), text[:60]))
    assert not bad, bad
