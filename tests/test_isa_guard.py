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
              "scale_yuv_kernel", "yuv2rgb_kernel", "scale_yuvu_rgb_kernel", "scale_yuvu_planes_kernel"):
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
    # Round 4: scale_yuv2s_blk_kernel uses the same S2_SAT2 / S2_JOIN construction (bytes 0 and 1 only); bit-exact on the GPU in both
    # forms at every launch size (profiles/r04d_block_form_one_frame_per_launch.txt, gpurun r04b: 1391 passed).
    # scale_yuvg_blk_rgb_kernel: the band walker's colour stage, same construction (g_sat_pk_u8_i16 -> v_perm_b32 of bytes 0 and 1); bit-exact
    # on the GPU (gpurun r04m / r04o: tests/test_parity_generic_walker.py in both forms, 518 passed).
    # rotate_mt_kernel: v_cvt_pk_u8_f32 is USED for its merge — it converts under MODE.fp_round (toward zero there) and drops the byte into
    # byte S1 of S2, the output dword being assembled (k_transform.hip rot_put_u8); the first of a dword's four starts from 0.  On the GPU:
    # tools/ubench/rot_probe.hip (2^24 blends against the 64-bit integer form, 0 differences), tests -k rotate 168 passed, 1500 fuzz cases
    # (profiles/r04_rotate.txt).
    # scale_yuvu_rgb_kernel (the quad-lane walker): the headline's colour stage (U_SAT2 -> U_JOIN: bytes 0 and 1 only); bit-exact on the GPU
    # (gpurun r04r / r04t: tools/probe/probe_yuvu.py, 162 cases; tests/test_parity_quad_walker.py).
    verified = {"v_sat_pk_u8_i16": ("scale_yuv2s_kernel", "scale_yuv2s_blk_kernel", "scale_yuvg_rgb_kernel", "scale_yuvg_blk_rgb_kernel", "scale_yuvu_rgb_kernel"),
                "v_cvt_pk_u8_f32": ("rotate_mt_kernel",)}
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
    strip = ("scale_yuv2s_kernel", "scale_yuv2s_blk_kernel", "scale_yuv2s_np_kernel", "scale_yuv2p_kernel", "scale_yuv2px_kernel", "scale_yuv1x2_kernel", "scale_yuv3x1_kernel", "scale_yuv3r_kernel", "scale_yuv3x2_kernel", "scale_yuv32r_kernel", "scale_yuv4r_kernel", "scale_yuv4x1_kernel",
             "scale_yuvg_blk_rgb_kernel", "scale_yuvg_blk_planes_kernel", "scale_yuvu_rgb_kernel", "scale_yuvu_planes_kernel", "scale_rgb2s_kernel", "scale_rgb2h_kernel", "scale_rgb2y_kernel", "rgb2yuv420s_kernel", "smooth121_kernel",
             "scale19_kernel", "scale19_unit_kernel", "scale19_unit64_kernel", "unit_rgb_kernel")          # (round 6: handing the frame table to its device functions by reference copied 1.5 KB into every lane's scratch — seven times slower, profiles/r06_scale19_history.txt r06l)
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
    for k in ("scale_yuv2s_kernel", "scale_yuv2s_blk_kernel", "scale_yuv2p_kernel", "scale_yuv3r_kernel", "scale_yuv32r_kernel", "rgb2yuv420s_kernel", "yuv2rgb_kernel",
              "flip_direct_kernel", "median3x3s_kernel"):
        plain, nt = total(k)
        assert nt > 0, (k, plain, nt)
    for k in ("scale_yuvg_", "scale_yuvu_", "rotate_lds_kernel"):
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


# ---- an ISA-level hazard lint (round 4): the rules the compiler cannot apply inside an inline-asm string,
#      checked on the MACHINE CODE of every shipped code object instead of on the asm strings' text --------------------------------------
def _lint():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_hazard_lint_finds_nothing_in_the_shipped_library(disassembly):
    """rule A: no VMEM instruction reads an SGPR a VALU instruction wrote fewer than 5 wait states earlier; rule B: no VALU write of the
    data VGPRs right behind a store of more than 64 bits (k_scale_yuv2x.hip's global_store_dwordx3 is issued from an asm string) — in
    EVERY function of every gfx950 code object of libgmat_hip.so, whatever source construct made the instruction"""
    lint = _lint()
    bad, nvmem = [], 0
    for t in disassembly:
        bad += lint.lint(t)
        nvmem += len(re.findall(r"^\s+(?:buffer|global|flat|scratch)_(?:load|store)", t, re.M))
    assert nvmem > 5000, nvmem                          # not vacuous: the scan saw the library's memory instructions
    assert not bad, bad[:6]


def test_hazard_lint_rules_on_known_sequences():
    """the lint itself, on hand-written disassembly: the wait-state count (s_nop N = N + 1), SGPR ranges, the carry-out destinations"""
    lint = _lint().lint
    head = "0000000000001000 <k>:\n"
    bad = lint(head + "\tv_readfirstlane_b32 s4, v1 // 0\n\tv_mov_b32_e32 v2, v3 //\n\tbuffer_store_dword v0, v1, s[4:7], s9 offen //\n")
    assert len(bad) == 1 and bad[0][1] == "A"
    ok = lint(head + "\tv_readfirstlane_b32 s4, v1 //\n\ts_nop 4 //\n\tbuffer_store_dword v0, v1, s[4:7], s9 offen //\n")
    assert not ok
    ok = lint(head + "\tv_readfirstlane_b32 s4, v1 //\n" + "\tv_mov_b32_e32 v2, v3 //\n" * 5 + "\tglobal_load_dword v0, v1, s[4:5] //\n")
    assert not ok
    bad = lint(head + "\tv_readfirstlane_b32 s4, v1 //\n" + "\tv_mov_b32_e32 v2, v3 //\n" * 4 + "\tglobal_load_dword v0, v1, s[4:5] //\n")
    assert len(bad) == 1                                # four instructions between = four wait states: one short
    bad = lint(head + "\tv_add_co_u32_e64 v1, s[10:11], v2, v3 //\n\tglobal_store_dword v0, v1, s[10:11] //\n")
    assert len(bad) == 1                                # the carry-out of a VOP3 add is an SGPR pair a VALU wrote
    ok = lint(head + "\tv_readfirstlane_b32 s4, v1 //\n\tglobal_store_dword v0, v1, s[8:9] //\n")
    assert not ok                                       # another register
    bad = lint(head + "\tglobal_store_dwordx3 v0, v[4:6], off //\n\tv_mov_b32_e32 v5, v9 //\n")
    assert len(bad) == 1 and bad[0][1] == "B"
    ok = lint(head + "\tglobal_store_dwordx3 v0, v[4:6], off //\n\tv_mov_b32_e32 v7, v9 //\n")
    assert not ok
    ok = lint(head + "\tglobal_store_dwordx2 v0, v[4:5], off //\n\tv_mov_b32_e32 v5, v9 //\n")
    assert not ok                                       # 64 bits: no hazard
    bad = lint(head + "\tbuffer_store_dwordx4 v[8:11], v1, s[4:7], 0 offen //\n\tv_perm_b32 v10, v1, v2, v3 //\n")
    assert len(bad) == 1


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_hazard_lint_catches_the_walkers_store_without_its_guard(tmp_path):
    """Round 3's fault reproduced on the machine code: the band walker's inline-asm store with its `s_nop 4` taken out, compiled as the
    library compiles it — in today's layout the planar-chroma RGB instances reload the buffer resource's words from a spill lane
    (v_readlane_b32 s23) directly in front of the buffer_store that reads them: rule A, 0 wait states.  (The pre-fix source AS
    COMMITTED, f1442d4^, lints clean: its layout happened to keep the distance — 'dormant', FINDINGS.md R3-walker-bands — which is
    exactly why a source-string guard was not enough and the shipped code objects are linted on every run.)"""
    src = open(os.path.join(ROOT, "gmat_amd", "csrc", "k_scale_yuvg.hip")).read()
    assert src.count('"s_nop 4\\n\\t') == 1
    bare = tmp_path / "k_scale_yuvg_noguard.hip"
    bare.write_text(src.replace('"s_nop 4\\n\\t', '"'))
    obj = tmp_path / "noguard.o"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "--offload-arch=gfx950",
                        "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gmat_amd", "csrc"),
                        "--cuda-device-only", "-c", str(bare), "-o", str(obj)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    bad = _lint().lint_file(str(obj))
    assert bad and all(b[1] == "A" and "scale_yuvg" in b[0] and "buffer_store_dword" in b[2] for b in bad), bad[:4]
