#!/bin/bash
# A/B of the NTT pass kernels on one MI355X box (through gpurun): generated assembly (default) against the hipcc pass.
#   gpurun --timeout 600 -- 'bash tools/ntt_ab.sh > gpurun_out/ntt_ab.txt 2>&1'
set -o pipefail
for lg in 24 20 16; do
  timeout -k 10 120 python3 tools/ntt_probe.py $lg 6 || exit 1
  GH_NTT_ASM=0 timeout -k 10 120 python3 tools/ntt_probe.py $lg 6 || exit 1
done
timeout -k 10 120 python3 tools/ntt_probe.py 14 10 mnt6753_fr || exit 1
GH_NTT_ASM=0 timeout -k 10 120 python3 tools/ntt_probe.py 14 10 mnt6753_fr
