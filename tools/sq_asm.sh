#!/bin/bash
# SQ counters of the G1 accumulation kernel (assembly and hipcc builds), one --pmc pass each:  bash tools/sq_asm.sh OUTDIR
set -o pipefail
OUT=${1:-gpurun_out/sq_asm}
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/acc_probe.py"
for v in 1 0; do
  export GH_ACC_ASM=$v
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv \
      -d "$R/$OUT/a_asm$v" -o run -- $P mnt4753_g1 20 1 1 2 nocheck > "$R/$OUT/a_asm$v.log" 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU --output-format csv \
      -d "$R/$OUT/b_asm$v" -o run -- $P mnt4753_g1 20 1 1 2 nocheck > "$R/$OUT/b_asm$v.log" 2>&1 || echo "pass b failed for asm=$v"
done
cd "$R"
for v in 1 0; do
  for p in a b; do
    f=$(ls $OUT/${p}_asm$v/*/run_counter_collection.csv $OUT/${p}_asm$v/run_counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 tools/pmc_table.py "$f" acc_g1 accumulate_xyzz | tail -1
  done
done > "$OUT/summary.txt"
cat "$OUT/summary.txt"
