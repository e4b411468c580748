#!/bin/bash
# tools/ubsan_emu.sh <out dir> [fuzz cases] — sanitizers run on the CPU build only (no GPU ASan / XNACK on the pool): every kernel source and the host
# code, compiled against the CPU emulation of HIP (tests/hipemu) under clang's UndefinedBehaviorSanitizer, then the six differential fuzzers over it.
#   * -fsanitize=undefined with the MINIMAL runtime (the ROCm clang ships no other): reports are `ubsan: <kind> by <pc>` on descriptor 2, symbolised
#     below with llvm-symbolizer; recoverable, so one run lists every site;
#   * off: alignment (the kernels' unaligned vector loads are memcpy on the CPU), vptr / function (no RTTI in the kernels), shift-base — left shifts
#     of negative values and into the sign bit are two's-complement arithmetic here by design (C++20's rule; gfx950 and x86 agree), 50 sites;
#   * ASan is not usable on this build: the emulation runs a workgroup's threads as ucontext fibers; its guard-page allocator (hipemu.cpp) is what
#     catches reads and writes past a device allocation instead.
# Round 4 result (profiles/r04_ubsan.txt): 1 200 fuzz cases + the CPU suite (6 234 tests): no signed overflow, no out-of-range shift exponent, no
# division by zero, no bad bool / enum load, no out-of-bounds array index in any kernel or in the host code; ONE report, a pointer-arithmetic check in
# vrgba64_kernel's `ly[(size_t) r0 * lumW + x]` (k_scale16.hip) whose operands were then instrumented and are in range (r0, r1, x >= 0, ly != NULL).
# The test suite can use the same library: GMAT_TEST_EMU_LIBRARY=<out dir>/libgmat_hip_emu.so GMAT_SANITIZER_LOG=<prefix> python -m pytest tests -m "not gpu".
set -e
OUT=${1:?usage: ubsan_emu.sh <out dir> [fuzz cases]}; N=${2:-150}
R=$(cd $(dirname $0)/.. && pwd)
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_minimal-x86_64.a | head -1)
mkdir -p $OUT
sed "s#build/#$OUT/#g; s#@mkdir -p build#@mkdir -p $OUT#; s#CXXFLAGS := -O1 -g#CXXFLAGS := -O1 -g -fsanitize=undefined -fsanitize-minimal-runtime -fno-sanitize=alignment,vptr,function,shift-base#; s#\$(CXX) -shared -o \$@ \$(OBJS)#\$(CXX) -shared -o \$@ \$(OBJS) -Wl,--whole-archive $RT -Wl,--no-whole-archive#" $R/tests/hipemu/Makefile > $OUT/Makefile
( cd $R/tests/hipemu && make -f $OUT/Makefile -j8 > $OUT/make.log 2>&1 )
cat > $OUT/run.py <<PY
import os, sys, runpy, ctypes
os.environ["GMAT_TEST_EMU_LIBRARY"] = "$OUT/libgmat_hip_emu.so"
ctypes.CDLL("$OUT/libgmat_hip_emu.so")
for l in open("/proc/self/maps"):
    if "libgmat_hip_emu.so" in l and "r-xp" in l:
        print("BASE", l.split("-")[0], l.split()[2], file=sys.stderr); break
sys.argv = [sys.argv[1]] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
for f in fuzz_parity fuzz_strip fuzz_walker fuzz_yuvopts fuzz_unit fuzz_transforms fuzz_filters; do
  python $OUT/run.py $R/tests/fuzz/$f.py $N 4321 2> $OUT/err_$f.txt | tail -1 | sed "s/^/$f: /"
done
python3 - $OUT <<'PY'
import subprocess, re, glob, collections, sys
out = sys.argv[1]; res = collections.Counter()
for f in glob.glob(out + "/err_*.txt"):
    l = open(f).read().split("\n")
    base = [x for x in l if x.startswith("BASE")][0].split()
    b, off = int(base[1], 16), int(base[2], 16)
    for k, pc in set((m.group(1), int(m.group(2), 16)) for m in (re.match(r"ubsan: ([a-z-]+) by 0x([0-9a-f]+)", x) for x in l) if m):
        o = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + out + "/libgmat_hip_emu.so", "--inlines", hex(pc - b + off)],
                           capture_output=True, text=True).stdout.strip().split("\n")
        res[(k, o[0][:70], o[1] if len(o) > 1 else "?")] += 1
for (k, fn, loc), n in sorted(res.items(), key=lambda t: t[0][2]):
    print(k, "|", fn, "|", loc)
print("ubsan sites:", len(res))
PY
