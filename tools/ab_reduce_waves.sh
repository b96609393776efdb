#!/bin/bash
# A/B on one box: G2 bucket reduction inside a batch on its 512-register build (GH_REDUCE_WAVES=1) or its 256-register build (default)
for rep in 1 2; do
  for s in 1 0; do
    echo "== GH_REDUCE_WAVES=$s (0 = default)"
    if [ $s = 0 ]; then unset GH_REDUCE_WAVES; else export GH_REDUCE_WAVES=$s; fi
    timeout -k 10 200 python3 tools/g2_probe.py mnt4753_g2 20 1 0 4 2>&1 | tail -1
    timeout -k 10 200 python3 tools/g2_probe.py mnt6753_g2 19 1 0 4 2>&1 | tail -1
  done
done
