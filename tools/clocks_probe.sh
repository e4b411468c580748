#!/bin/bash
# tools/clocks_probe.sh — runs ON the GPU box: samples rocm-smi (clocks, power, temperature) while the headline case runs for a few seconds:
# is the launch-to-launch spread (p10 -> p50: 3 %) the clocks moving under a power limit?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
( X2BENCH_VERIFY=0 X2BENCH_SETS=8 tools/bin/x2bench 32 3000 "nv12 4K->1080p rgb24 bicubic" > /tmp/cp_x2.txt 2>&1 ) &
PID=$!
sleep 0.5
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)|Performance" | tr -s ' ' | tr '\n' ';' | cut -c1-600
  echo
  sleep 0.15
done
wait $PID
cut -c1-160 /tmp/cp_x2.txt
echo "== idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' ' | tr '\n' ';' | cut -c1-400; echo
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | grep -vE "^=|^$" | head -40
