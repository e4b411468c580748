#!/bin/bash
mkdir -p gpurun_out/land2
for c in "land: rgb24 4K->1080p nv12" "land: nv12 4K->720p rgb24" "land: nv12 1080p->360p"; do
  tools/bin/x2bench 32 10 "$c" 2>&1 | grep -v "^#"
done | tee gpurun_out/land2/x2.txt
