#!/bin/bash
# tools/isa.sh <file.hip> <mangled-name substring> [extra hipcc flags]: device assembly of one kernel -> /tmp/isa/<name>.s and a summary
# (VGPRs, scratch, instruction mix of the whole kernel).  Build-container tool; no GPU needed.
F=$1; K=$2; shift 2
mkdir -p /tmp/isa; B=$(basename $F .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$(dirname $0)/../include -I$(dirname $0)/../gmat_amd/csrc --cuda-device-only -S "$@" -o /tmp/isa/$B.s $F 2>/dev/null
python3 - /tmp/isa/$B.s "$K" <<'PY'
import re, sys, collections
txt = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\s*\.end_amdhsa_kernel", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if key not in name: continue
    ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    top = ", ".join("%s %d" % kv for kv in c.most_common(14))
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body); sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
    print(name[:110]); print("   vgpr", vg and vg.group(1), "scratch", sc and sc.group(1), "instructions", len(ins), "VALU", valu, "scratch_ops", sum(v for k, v in c.items() if "scratch" in k))
    print("   ", top)
PY
