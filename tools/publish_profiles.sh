#!/bin/bash
# Copies the summaries of one tools/collect_profiles.sh run (merged back under gpurun_out/) into profiles/, named per round:
#   bash tools/publish_profiles.sh gpurun_out/fin1/prof r02
set -e
SRC=$1; R=$2
cp "$SRC/stats/run_kernel_stats.csv" "profiles/${R}_rocprofv3_kernel_stats_bench.csv"
cp "$SRC/pmc_traffic.json" "profiles/${R}_pmc_traffic.json"
for k in g1 g2 g2m6 ntt; do for c in FETCH_SIZE WRITE_SIZE; do
  python3 tools/pmc_table.py "$SRC/${k}_$c/run_counter_collection.csv" msm_accumulate aff_round gh_asm aff_inv ntt_ > "profiles/${R}_pmc_${k}_$c.txt"
done; done
for k in g1 g2 g2m6; do
  [ -f "$SRC/sq_$k/run_counter_collection.csv" ] && python3 tools/pmc_table.py "$SRC/sq_$k/run_counter_collection.csv" msm_accumulate aff_round gh_asm msm_wave_reduce > "profiles/${R}_sq_$k.txt"
done
ls -la profiles | grep "${R}_"
