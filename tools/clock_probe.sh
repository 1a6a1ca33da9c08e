#!/bin/bash
# Shader clock and power while a workload runs (GPU box): bash tools/clock_probe.sh "<python command>"
# Samples rocm-smi every 0.5 s while the command runs in the background (started as a child, never exec'd from a GPU process).
set -u
CMD="$1"
$CMD > /tmp/clock_probe_out.txt 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket Power\|mclk" | head -4 | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
tail -2 /tmp/clock_probe_out.txt
