#!/bin/bash
# Partial shift tables at 2^24 pairs (MNT4-753 G1, c = 21): rows of the table against time per MSM.  bash tools/partial_table_sweep.sh <out>
OUT=${1:-gpurun_out/partial_tables.txt}
: > "$OUT"
python3 tools/acc_probe.py mnt4753_g1 24 0 3 2 nocheck 2>&1 | tail -2 | sed 's/^/[no table] /' >> "$OUT"
for R in 2 4 6 8 12 18 0; do
  GH_TABLE_ROWS=$R PROBE_C=21 python3 tools/acc_probe.py mnt4753_g1 24 1 3 2 nocheck 2>&1 | tail -2 >> "$OUT"
done
cat "$OUT"
