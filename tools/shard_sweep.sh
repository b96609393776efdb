#!/bin/bash
# Shard-size / table-window sweep behind DESIGN.md section 6 (config 4's predicted 1 / 2 / 4 / 8-GPU curve): one MSM and a
# pipelined batch per (curve, log2 pairs, table window).  Usage (through gpurun): bash tools/shard_sweep.sh gpurun_out/xx/sweep.txt
OUT=${1:-gpurun_out/shard_sweep.txt}
: > "$OUT"
run() { PROBE_C=$3 python3 tools/acc_probe.py $1 $2 1 4 2 nocheck 2>&1 | grep -E "single|batch" | tail -2 | sed "s/^/[PROBE_C=$3] /" >> "$OUT"; }
for c in 18 20 21; do run mnt6753_g1 19 $c; done
for c in 20 21; do run mnt6753_g1 20 $c; done
run mnt6753_g1 21 21
run mnt6753_g1 22 21
for c in 16 17 18 19; do run mnt6753_g2 19 $c; done
for c in 18 19; do run mnt6753_g2 20 $c; done
run mnt6753_g2 21 19
for c in 19 21; do run mnt6753_g2 22 $c; done
cat "$OUT"
