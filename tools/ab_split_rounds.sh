#!/bin/bash
# A/B on one box: affine rounds issued in one piece (GH_AFF_SPLIT=0) or as two halves on two streams (default)
for rep in 1 2; do
  for s in 0 1; do
    echo "== GH_AFF_SPLIT=$s"
    GH_AFF_SPLIT=$s timeout -k 10 200 python3 tools/g2_probe.py mnt4753_g2 20 2 0 4 2>&1 | tail -2
    GH_AFF_SPLIT=$s timeout -k 10 200 python3 tools/g2_probe.py mnt6753_g2 19 2 0 4 2>&1 | tail -2
  done
done
