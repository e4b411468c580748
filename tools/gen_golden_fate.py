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


def main():
    doc = {
        "_source": "ffmpeg-gpu/tests/ref/fate/* and tests/ref/pixfmt/* of the reference tree (golden values only)",
        "framecrc": {n: framecrc(n) for n in ("sws-yuv-range", "sws-yuv-colorspace", "filter-scalechroma", "filter-colorlevels",
                                              "filter-transpose")},
        "pixfmt_md5": {n: md5ref(n) for n in ("rgb24", "bgr24", "yuv420p")},
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "fate_refs.json")
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    sys.exit(main())
