#!/bin/bash
# tools/build_ref_avfilter.sh <out dir> [<lib dir> <lib name>] — BUILD CONTAINER ONLY (needs /root/reference; nothing of it travels).
# The reference's REAL libavfilter + libavutil + libswscale (out of tree, portable C, --disable-everything + the CPU filters the parity
# target names) drive this repository's reference-side sources:
#     integration/hwcontext_hip.c      fills libavutil's ONE open symbol, ff_hwcontext_type_cuda (hwcontext.c:36-38)
#     integration/vf_gmat_hip.c        the seven GPU filters          integration/vf_hwupload_hip.c   upload through the pinned ring
#     integration/swscale_hip_adapter.c + the library                 libswscale's nine open symbols (tools/build_ref_swscale.sh)
# with the library under test (default: the CPU-emulated build).  tests/c/avfilter_graph_caller.c builds two graphs with libavfilter's public
# API — buffer -> hwupload_hip -> <GPU filters> -> hwdownload -> buffersink and buffer -> <the CPU filters> -> buffersink — and compares bytes.
# Two things are done to the GENERATED files in <out dir> (never to the reference): config.h says CONFIG_CUDA 1 so that hwcontext.c lists the
# slot (what a `--enable-hip` switch of configure would do), and a two-line cuda.h supplies the CUcontext / CUstream handle types that
# libavutil/hwcontext_cuda.h includes.  Test infrastructure, not an oracle/_ref build.
set -e
REF=/root/reference/ffmpeg-gpu
OUT=${1:?usage: build_ref_avfilter.sh <out dir> [<lib dir> <lib name>]}
R=$(cd $(dirname $0)/.. && pwd)
LIBDIR=${2:-$R/tests/hipemu/build}; LIBNAME=${3:-gmat_hip_emu}
[ -x $REF/configure ] || { echo "reference tree not present"; exit 77; }
mkdir -p $OUT/shim && cd $OUT
printf 'typedef struct CUctx_st *CUcontext;\ntypedef struct CUstream_st *CUstream;\n' > shim/cuda.h
if [ ! -f libavfilter/libavfilter.a ] || [ ! -f libswscale/libswscale.a ] || [ ! -f libavutil/libavutil.a ]; then
  bash $REF/configure --disable-asm --disable-doc --disable-autodetect --disable-network --disable-everything --disable-programs \
       --enable-filter=scale,format,hwdownload,hwupload,transpose,hflip,vflip,crop,rotate,convolution,median,null \
       --extra-cflags=-I$OUT/shim > configure.log 2>&1
  sed -i 's/^#define CONFIG_CUDA 0$/#define CONFIG_CUDA 1/' config.h
  make -j8 libavfilter/libavfilter.a libswscale/libswscale.a libavutil/libavutil.a > make.log 2>&1
fi
# what the static libraries leave open is exactly what this repository supplies
nm -u libavutil/libavutil.a libswscale/libswscale.a libavfilter/libavfilter.a | awk '{print $2}' | grep -E '_cuda$' | sort -u > open_symbols.txt
INC="-I$OUT -I$REF -I$OUT/shim -I$R/include"
CF="-std=c11 -O1 -Wall -DHAVE_AV_CONFIG_H -D_ISOC11_SOURCE -D_DEFAULT_SOURCE -D_XOPEN_SOURCE=600"
gcc $CF $INC -I$REF/libavutil -c $R/integration/hwcontext_hip.c -o hwcontext_hip.o
gcc $CF $INC -I$REF/libavfilter -c $R/integration/vf_gmat_hip.c -o vf_gmat_hip.o
gcc $CF $INC -I$REF/libavfilter -c $R/integration/vf_hwupload_hip.c -o vf_hwupload_hip.o
gcc $CF $INC -c $R/integration/swscale_hip_adapter.c -o adapter.o
gcc -std=c11 -O1 -Wall -D_DEFAULT_SOURCE $INC -c $R/tests/c/avfilter_graph_caller.c -o graph_caller.o
gcc graph_caller.o vf_gmat_hip.o vf_hwupload_hip.o hwcontext_hip.o adapter.o libavfilter/libavfilter.a libswscale/libswscale.a libavutil/libavutil.a \
    -L$LIBDIR -l$LIBNAME -Wl,-rpath,$LIBDIR -lm -lpthread -o avfilter_graph_caller
echo "built $OUT/avfilter_graph_caller"
