#!/bin/bash
# Reproduces the rocprofv3 evidence under profiles/ on an MI355X box (run from the repo root through gpurun):
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh gpurun_out/prof_rNN'
# Kernel stats and counters are collected in separate runs (a --pmc run carries only --kernel-trace), the
# program itself follows `--` (no env / bash -c hop: the profiler initialises the GPU before the program starts).
set -o pipefail
OUT=${1:-gpurun_out/prof}
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# chain keys (distinct bases, as bench.py's): a tiled key is a key of equal bases, which the library now adds up front
P="python3 $R/tools/g2_probe.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o run -- python3 "$R/bench.py" --no-cpu-baseline --no-2p24 > "$R/$OUT/stats.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$OUT/g1_$c" -o run -- $P mnt4753_g1 20 2 > "$R/$OUT/g1_$c.log" 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$OUT/g2_$c" -o run -- $P mnt4753_g2 20 2 > "$R/$OUT/g2_$c.log" 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$OUT/g2m6_$c" -o run -- $P mnt6753_g2 19 2 > "$R/$OUT/g2m6_$c.log" 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$OUT/ntt_$c" -o run -- python3 "$R/tools/prof_run.py" ntt 24 3 > "$R/$OUT/ntt_$c.log" 2>&1 || exit 1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv \
    -d "$R/$OUT/sq_g1" -o run -- $P mnt4753_g1 20 2 > "$R/$OUT/sq_g1.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv \
    -d "$R/$OUT/sq_g2" -o run -- $P mnt4753_g2 20 2 > "$R/$OUT/sq_g2.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv \
    -d "$R/$OUT/sq_g2m6" -o run -- $P mnt6753_g2 19 2 > "$R/$OUT/sq_g2m6.log" 2>&1 || exit 1
cd "$R"
python3 tools/make_traffic_json.py "$OUT/pmc_traffic.json" \
  "mnt4753_g1_2p20:$OUT/g1_FETCH_SIZE/run_counter_collection.csv:$OUT/g1_WRITE_SIZE/run_counter_collection.csv:21:XYZZ mixed additions (madd-2008-s, 8 M + 2 S, Y3 as one dual product)" \
  "mnt6753_g2_2p19:$OUT/g2m6_FETCH_SIZE/run_counter_collection.csv:$OUT/g2m6_WRITE_SIZE/run_counter_collection.csv:19:affine rounds (asmgen/g2_rounds.py) + projective finish" \
  "mnt4753_g2_2p20:$OUT/g2_FETCH_SIZE/run_counter_collection.csv:$OUT/g2_WRITE_SIZE/run_counter_collection.csv:19:affine rounds (asmgen/g2_rounds.py) + projective finish" \
  "ntt_2p24:$OUT/ntt_FETCH_SIZE/run_counter_collection.csv:$OUT/ntt_WRITE_SIZE/run_counter_collection.csv:0:-" > "$OUT/pmc_traffic.log" 2>&1
ls "$R/$OUT"
