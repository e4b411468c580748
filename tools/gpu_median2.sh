#!/bin/bash
for r in 2 4 6 8 10; do echo "== strip, GMAT_STRIP_ROWS=$r"; GMAT_STRIP_ROWS=$r tools/bin/x2bench 1 20 "op: median" | grep -v "^#"; done 2>&1 | tee gpurun_out/median/x2b.txt
