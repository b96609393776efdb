#!/bin/bash
# A/B of library variants on one box: bash tools/ab_variants.sh v0 v1 ...  (ginger-lib_amd/libginger_hip_<tag>.so, built beside the shipped one)
set -e
L=ginger-lib_amd
cp $L/libginger_hip.so /tmp/shipped.so
for round in 1 2; do
  for t in "$@"; do
    cp $L/libginger_hip_$t.so $L/libginger_hip.so
    echo "== $t"; python3 tools/acc_probe.py mnt4753_g1 20 1 10 2 nocheck 2>&1 | grep -v precompute | tail -2 | cut -c1-140
  done
done
cp /tmp/shipped.so $L/libginger_hip.so
