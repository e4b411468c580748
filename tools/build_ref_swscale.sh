#!/bin/bash
# tools/build_ref_swscale.sh <out dir> [<lib dir> <lib name>] — BUILD CONTAINER ONLY (needs /root/reference; nothing of it travels).
# Builds the reference's libswscale.a + libavutil.a out of tree (configure into <out dir>, portable C,
# --disable-everything) and links tests/c/libswscale_core_caller.c with integration/swscale_hip_adapter.c and the library under test
# (default: the CPU-emulated build) IN PLACE OF the nine symbols the reference's libswscale/cuda objects define:
#     ff_sws_init_swscale_cuda ff_sws_free_swscale_cuda ff_swscale_cuda ff_yuv2rgb_init_tables_cuda      <- the adapter
#     yuv2rgb_cuda rgb2yuv_cuda yuv2yuv_cuda rgb24tobgr24_cuda rgb2rgb_init_cuda                          <- the library itself
# No stub of any of them is written.  The only stand-in is a two-line cuda.h for the CUcontext / CUstream handle types, which
# libswscale/swscale_internal.h includes unconditionally (defect 9).  Test infrastructure: this is NOT an
# oracle/_ref build (the oracle stays the restatement) — it checks that the reference's real core, not a
# hand-written imitation of its SwsContext, can drive the back-end.
set -e
REF=/root/reference/ffmpeg-gpu
OUT=${1:?usage: build_ref_swscale.sh <out dir> [<lib dir> <lib name>]}
R=$(cd $(dirname $0)/.. && pwd)
LIBDIR=${2:-$R/tests/hipemu/build}; LIBNAME=${3:-gmat_hip_emu}
[ -x $REF/configure ] || { echo "reference tree not present"; exit 77; }
mkdir -p $OUT/shim && cd $OUT
printf 'typedef struct CUctx_st *CUcontext;\ntypedef struct CUstream_st *CUstream;\n' > shim/cuda.h
if [ ! -f libswscale/libswscale.a ] || [ ! -f libavutil/libavutil.a ]; then
  bash $REF/configure --disable-asm --disable-doc --disable-autodetect --disable-network --disable-everything --disable-programs \
       --extra-cflags=-I$OUT/shim > configure.log 2>&1
  make -j8 libswscale/libswscale.a libavutil/libavutil.a > make.log 2>&1
fi
# exactly the nine symbols are open, and they are the ones named above
nm -u libswscale/libswscale.a | awk '{print $2}' | grep -E '_cuda$' | sort -u > open_symbols.txt
INC="-I$OUT -I$REF -I$OUT/shim -I$R/include"
gcc -std=c11 -O1 -Wall -DHAVE_AV_CONFIG_H -D_ISOC11_SOURCE -D_DEFAULT_SOURCE $INC -c $R/integration/swscale_hip_adapter.c -o adapter.o
gcc -std=c11 -O1 -Wall $INC -c $R/tests/c/libswscale_core_caller.c -o caller.o
gcc caller.o adapter.o libswscale/libswscale.a libavutil/libavutil.a -L$LIBDIR -l$LIBNAME -Wl,-rpath,$LIBDIR -lm -lpthread -o libswscale_core_caller
echo "built $OUT/libswscale_core_caller"
