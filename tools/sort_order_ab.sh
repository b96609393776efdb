#!/bin/bash
# A/B of the bucket lists' order (GH_SORT_ORDERED=0: second sweep of msm_bin_sort_kernel without its barriers) on one box:
#   bash tools/sort_order_ab.sh > gpurun_out/<file>
for shape in "mnt4753_g1 20 1" "mnt4753_g1 20 0" "mnt4753_g1 22 1" "mnt4753_g1 22 0" "mnt4753_g1 24 0" "mnt4753_g2 20 1" "mnt6753_g2 19 1"; do
  for o in 1 0; do
    GH_SORT_ORDERED=$o python3 tools/acc_probe.py $shape 4 3 nocheck 2>&1 | grep -v precompute | tail -3
  done
done
