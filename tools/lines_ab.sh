#!/bin/bash
# tools/lines_ab.sh <nf> <launches> — runs ON the GPU box: the "thumb:" rows of x2bench (and two up-scales) on the lines form and on the tiled
# kernel with every walker switched off, beside the default rule: the A/B behind the rule in gsws.cpp lines_context()
NF=${1:-32}; L=${2:-20}
OFF="GMAT_SCALE_NO_STRIP=1 GMAT_SCALE_NO_GENERIC_WALKER=1 GMAT_SCALE_NO_QUAD_WALKER=1"
echo "== default rule, $NF frames a launch"; tools/bin/x2bench $NF $L "thumb:" 2>&1
echo "== lines forced (walkers off, GMAT_LINES=2), $NF frames a launch"
env $OFF GMAT_LINES=2 tools/bin/x2bench $NF $L "thumb:" 2>&1
env $OFF GMAT_LINES=2 tools/bin/x2bench $NF $L "any: up nv12 720p->1080p" 2>&1
env $OFF GMAT_LINES=2 tools/bin/x2bench $NF $L "any: short nv12 1440p->1080p" 2>&1
echo "== tiled forced (walkers off, GMAT_LINES=0), $NF frames a launch"
env $OFF GMAT_LINES=0 timeout 300 tools/bin/x2bench $NF $(( L / 4 + 1 )) "thumb:" 2>&1
env $OFF GMAT_LINES=0 tools/bin/x2bench $NF $L "any: up nv12 720p->1080p" 2>&1
env $OFF GMAT_LINES=0 tools/bin/x2bench $NF $L "any: short nv12 1440p->1080p" 2>&1
