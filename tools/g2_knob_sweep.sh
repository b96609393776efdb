#!/bin/bash
# Knobs of the affine rounds on G2: scratch budget (chunks of buckets), points left per bucket for the
# projective finish, minimum batch per inversion.  Usage (through gpurun): bash tools/g2_knob_sweep.sh gpurun_out/xx/g2_knobs.txt
OUT=${1:-gpurun_out/g2_knobs.txt}
: > "$OUT"
run() { env "${@:3}" python3 tools/acc_probe.py $1 $2 1 3 2 nocheck 2>&1 | grep -E "single|batch" | tail -2 >> "$OUT"; }
for crv in "mnt4753_g2 20" "mnt6753_g2 19"; do
  run $crv GH_NOP=1
  run $crv GH_AFF_SCRATCH_GB=64
  run $crv GH_AFF_SCRATCH_GB=64 GH_AFF_LEFTOVER=2.5
  run $crv GH_AFF_SCRATCH_GB=64 GH_AFF_LEFTOVER=1.5
  run $crv GH_AFF_SCRATCH_GB=64 GH_AFF_LEFTOVER=2.5 GH_AFF_BMIN=4
  run $crv GH_AFF_SCRATCH_GB=64 GH_AFF_LEFTOVER=2.5 GH_AFF_BMIN=16
done
cat "$OUT"
