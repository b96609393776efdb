#!/bin/bash
# Reproduces the rocprofv3 evidence under profiles/ on an MI355X box (run from the repo root through gpurun):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh gpurun_out/prof_rNN'
# Kernel stats and counters are collected in separate runs (a --pmc run carries only --kernel-trace), the
# program itself follows `--` (no env / bash -c hop: the profiler initialises the GPU before the program starts).
set -o pipefail
OUT=${1:-gpurun_out/prof}
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o run -- python3 "$R/bench.py" --no-cpu-baseline > "$R/$OUT/stats.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/$OUT/fetch" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$OUT/fetch.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/$OUT/write" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$OUT/write.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv \
    -d "$R/$OUT/sq" -o run -- python3 "$R/tools/prof_run.py" msm 20 3 table > "$R/$OUT/sq.log" 2>&1 || exit 1
ls "$R/$OUT"/*/
