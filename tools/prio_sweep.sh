#!/bin/bash
# Stream-priority A/B for the pipelined MSM batch (one box, one process per setting): bash tools/prio_sweep.sh out.txt
OUT=${1:-gpurun_out/prio_sweep.txt}
: > "$OUT"
run() { env "$@" python3 tools/acc_probe.py mnt4753_g1 20 1 12 1 nocheck 2>&1 | grep -E "batch" | tail -1 >> "$OUT"; }
run GH_NOP=1
run GH_PRIO_ACC=normal
run GH_PRIO_ACC=normal GH_PRIO_RED=normal
run GH_PRIO_ACC=high GH_PRIO_RED=high
run GH_NOP=2
cat "$OUT"
