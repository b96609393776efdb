#!/bin/bash
# Samples clock and power of every visible GPU (rocm-smi) while a workload runs: bash tools/power_probe.sh <curve> <log_n> <batch>
#   gpurun --timeout 600 -- 'bash tools/power_probe.sh mnt4753_g1 20 400 > gpurun_out/power_probe.txt 2>&1'
C=${1:-mnt4753_g1}; L=${2:-20}; B=${3:-400}
python3 tools/g2_probe.py $C $L 1 0 $B > /tmp/pp_run.log 2>&1 &
PID=$!
for i in $(seq 1 14); do
  sleep 1.5
  echo "t=$i"; rocm-smi --showpower --showclocks 2>&1 | grep -i "sclk\|Package Power" | awk '{print $1, $(NF-1), $NF}' | paste - - | awk '$0 !~ /95Mhz/ {print}' | head -8
done
wait $PID
tail -2 /tmp/pp_run.log
