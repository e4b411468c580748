#!/usr/bin/env python3
"""Extract the reference's own FATE golden values for the pixel-transform path into
tests/golden/fate_refs.json (data only: frame checksums and file digests, no reference source).

Run in the build container (needs /root/reference); the JSON it writes is committed and is what
tests/test_oracle_fate.py checks the oracle against.  Sources, all under
/root/reference/ffmpeg-gpu/tests/ref:
  fate/sws-yuv-range       (recipe tests/fate/libswscale.mak:28-34)
  fate/sws-yuv-colorspace  (recipe tests/fate/libswscale.mak:20-26)
  fate/filter-scalechroma  (tests/fate/filter-video.mak:416-418)
  fate/filter-colorlevels  (tests/fate/filter-video.mak:423-424; colorlevels with default options is
                            the identity, vf_colorlevels.c:405-439 -> the checksums are those of
                            scale's yuv420p->rgb24 output)
  fate/filter-transpose    (tests/fate/filter-video.mak:297-298)
  pixfmt/{rgb24,bgr24,yuv420p}  (tests/fate-run.sh:446-456 pixfmt_conversion: md5 of the yuv444p file)
  fate/filter-{null,vflip,crop,...}      (tests/fate-run.sh:458-466 video_filter: md5 of the NUT stream of 5 raw frames;
                                          recipes tests/fate/filter-video.mak:391-441)
  fate/filter-pixfmts-{null,copy,hflip,vflip,crop,transpose,rotate,scale}   (tests/fate-run.sh:468-497 pixfmts: one md5 of
                                          a 1-frame NUT stream per pixel format; recipes filter-video.mak:543-610)
"""
import json, os, re, sys

REF = "/root/reference/ffmpeg-gpu/tests/ref"


def framecrc(name):
    out = []
    for line in open(os.path.join(REF, "fate", name)):
        if line.startswith("#"):
            continue
        f = [x.strip() for x in line.split(",")]
        out.append({"size": int(f[4]), "adler32": f[5]})
    return out


def md5ref(name):
    first = open(os.path.join(REF, "pixfmt", name)).readline().split()
    return first[0]


VIDEO_FILTER = ("null", "vflip", "crop", "crop_vflip", "vflip_crop", "vflip_vflip", "scale200", "scale500", "crop_scale")
PIXFMTS_FILTERS = ("null", "copy", "hflip", "vflip", "crop", "transpose", "rotate", "scale")
PIXFMTS = ("yuv420p", "nv12", "rgb24", "bgr24", "rgba", "bgra", "yuv444p", "p010le",
           "p016le", "yuv444p16le", "rgba64le", "bgra64le",        # the 19-bit path's destinations
           "yuv420p10le", "yuv420p16le")                           # planar high-depth 4:2:0 (swscale_cuda.c:34-44)


def nutmd5(name):
    out = {}
    for line in open(os.path.join(REF, "fate", name)):
        f = line.split()
        if len(f) == 2:
            out[f[0]] = f[1]
    return out


def main():
    doc = {
        "_source": "ffmpeg-gpu/tests/ref/fate/* and tests/ref/pixfmt/* of the reference tree (golden values only)",
        "framecrc": {n: framecrc(n) for n in ("sws-yuv-range", "sws-yuv-colorspace", "filter-scalechroma", "filter-colorlevels",
                                              "filter-transpose")},
        "pixfmt_md5": {n: md5ref(n) for n in ("rgb24", "bgr24", "yuv420p")},
        "nut_md5": {
            "video_filter": {n: nutmd5("filter-" + n)[n] for n in VIDEO_FILTER},
            "pixfmts": {f: {k: v for k, v in nutmd5("filter-pixfmts-" + f).items() if k in PIXFMTS} for f in PIXFMTS_FILTERS},
        },
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "fate_refs.json")
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    sys.exit(main())
