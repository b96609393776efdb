#!/bin/bash
# A/B of two builds of the assembly kernels on ONE box: bash tools/ab_hsaco.sh build/variant_old/gh_asm.hsaco build/variant_new/gh_asm.hsaco
A=$1; B=$2
for rep in 1 2; do
  for v in $A $B; do
    echo "== $v"
    GH_ASM_HSACO=$v timeout -k 10 200 python3 tools/g2_probe.py mnt4753_g1 20 2 0 10 2>&1 | tail -2
    GH_ASM_HSACO=$v timeout -k 10 200 python3 tools/g2_probe.py mnt4753_g2 20 1 0 3 2>&1 | tail -1
    GH_ASM_HSACO=$v timeout -k 10 200 python3 tools/g2_probe.py mnt6753_g2 19 1 0 3 2>&1 | tail -1
    GH_ASM_HSACO=$v timeout -k 10 100 python3 tools/ntt_probe.py 24 4 2>&1 | head -1
  done
done
