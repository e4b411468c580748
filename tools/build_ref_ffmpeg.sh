#!/bin/bash
# tools/build_ref_ffmpeg.sh <out dir> [<lib dir> <lib name>] — BUILD CONTAINER ONLY (needs /root/reference; nothing of it travels).
# The reference's own `ffmpeg` PROGRAM (fftools + libavfilter + libavdevice's lavfi + libavcodec / libavformat rawvideo, framecrc, framemd5 — configured
# out of tree, portable C, --disable-everything) with this repository's reference-side sources in its libraries:
#     libavutil.a     += integration/hwcontext_hip.c          (ff_hwcontext_type_cuda: `-init_hw_device cuda=gpu:0` creates a HIP device)
#     libavfilter.a   += integration/vf_gmat_hip.c, vf_hwupload_hip.c   (crop / flip / rotate / transpose / smooth / scale / format / hwupload _hip by NAME)
#     libswscale.a    += integration/swscale_hip_adapter.c    (+ the library: libswscale's nine open symbols)
# so that command lines a GMAT user types run as they are:  ffmpeg -i ... -vf hwupload_hip,scale_hip=w=1920:h=1080:format=rgb24,hwdownload ...
# Three edits, all to GENERATED files of <out dir> (never to the reference), each what a configure switch of an integrated tree would write:
#     config.h                     CONFIG_CUDA 1 (hwcontext.c lists the slot)
#     libavfilter/filter_list.c    the eight filters appended (the `extern const AVFilter ff_vf_*_hip;` lines of allfilters.c)
#     ffbuild/config.mak           -lnvcv_types -lcvcuda dropped from EXTRALIBS-swscale (libgmat_hip takes their place: --extra-libs)
# plus the two-line cuda.h of tools/build_ref_swscale.sh.  tests/test_ffmpeg_cli.py compares framecrc / framemd5 of GPU and CPU command lines.
set -e
REF=/root/reference/ffmpeg-gpu
OUT=${1:?usage: build_ref_ffmpeg.sh <out dir> [<lib dir> <lib name>]}
R=$(cd $(dirname $0)/.. && pwd)
LIBDIR=${2:-$R/tests/hipemu/build}; LIBNAME=${3:-gmat_hip_emu}
[ -x $REF/configure ] || { echo "reference tree not present"; exit 77; }
mkdir -p $OUT/shim && cd $OUT
printf 'typedef struct CUctx_st *CUcontext;\ntypedef struct CUstream_st *CUstream;\n' > shim/cuda.h
if [ ! -f libavfilter/libavfilter.a ]; then
  bash $REF/configure --disable-asm --disable-doc --disable-autodetect --disable-network --disable-everything \
       --enable-decoder=rawvideo,wrapped_avframe --enable-encoder=rawvideo,wrapped_avframe --enable-muxer=framecrc,framemd5,null,rawvideo \
       --enable-protocol=file,pipe,md5 --enable-indev=lavfi \
       --enable-filter=scale,format,hwdownload,hwupload,transpose,hflip,vflip,crop,rotate,convolution,median,null,testsrc2,testsrc,rgbtestsrc,yuvtestsrc,copy,trim,fps \
       --extra-cflags=-I$OUT/shim --extra-ldflags="-L$LIBDIR -Wl,-rpath,$LIBDIR" --extra-libs="-l$LIBNAME -lm" > configure.log 2>&1
  sed -i 's/^#define CONFIG_CUDA 0$/#define CONFIG_CUDA 1/' config.h
  sed -i 's/-lnvcv_types//g; s/-lcvcuda//g' ffbuild/config.mak
  python3 - <<'PY'
p = "libavfilter/filter_list.c"; s = open(p).read()
names = ["crop_hip", "flip_hip", "rotate_hip", "transpose_hip", "smooth_hip", "scale_hip", "format_hip", "hwupload_hip"]
s = "".join("extern const AVFilter ff_vf_%s;\n" % n for n in names) + s.replace("    &ff_asrc_abuffer,", "".join("    &ff_vf_%s,\n" % n for n in names) + "    &ff_asrc_abuffer,")
open(p, "w").write(s)
PY
  make -j8 libavfilter/libavfilter.a libswscale/libswscale.a libavutil/libavutil.a > make_libs.log 2>&1
fi
INC="-I$OUT -I$REF -I$OUT/shim -I$R/include"
CF="-std=c11 -O1 -Wall -DHAVE_AV_CONFIG_H -D_ISOC11_SOURCE -D_DEFAULT_SOURCE -D_XOPEN_SOURCE=600"
gcc $CF $INC -I$REF/libavutil -c $R/integration/hwcontext_hip.c -o hwcontext_hip.o
gcc $CF $INC -I$REF/libavfilter -c $R/integration/vf_gmat_hip.c -o vf_gmat_hip.o
gcc $CF $INC -I$REF/libavfilter -c $R/integration/vf_hwupload_hip.c -o vf_hwupload_hip.o
gcc $CF $INC -c $R/integration/swscale_hip_adapter.c -o adapter.o
ar r libavfilter/libavfilter.a vf_gmat_hip.o vf_hwupload_hip.o 2> /dev/null
ar r libavutil/libavutil.a hwcontext_hip.o 2> /dev/null
ar r libswscale/libswscale.a adapter.o 2> /dev/null
rm -f ffmpeg ffmpeg_g
make -j8 ffmpeg > make_ffmpeg.log 2>&1
echo "built $OUT/ffmpeg"
