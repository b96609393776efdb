#!/bin/bash
# SQ counters of the G2 round kernels, one --pmc pass each:  bash tools/sq_g2.sh OUTDIR [curve log_n]
set -o pipefail
OUT=${1:-gpurun_out/sq_g2}
CURVE=${2:-mnt4753_g2}
LOGN=${3:-20}
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/g2_probe.py $CURVE $LOGN 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv \
    -d "$R/$OUT/a" -o run -- $P > "$R/$OUT/a.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU --output-format csv \
    -d "$R/$OUT/b" -o run -- $P > "$R/$OUT/b.log" 2>&1 || echo "pass b failed"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_WR SQ_IFETCH --output-format csv \
    -d "$R/$OUT/c" -o run -- $P > "$R/$OUT/c.log" 2>&1 || echo "pass c failed"
cd "$R"
for p in a b c; do
  f=$(ls $OUT/$p/*/run_counter_collection.csv $OUT/$p/run_counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    for k in aff_f2_fwd_r0 aff_f2_bwd_r0 aff_f2_bwd_rn aff_f3_fwd_r0 aff_f3_bwd_r0 aff_f3_bwd_rn; do
      python3 tools/pmc_table.py "$f" $k $k | head -1
    done
  fi
done > "$OUT/summary.txt"
cat "$OUT/summary.txt"
