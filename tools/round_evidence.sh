#!/bin/bash
# One gpurun call that produces the round's measured evidence (run from the repo root on the GPU box):
#   bash tools/round_evidence.sh gpurun_out/<dir> rNN
# 1. rocprofv3 kernel stats of the bench + PMC passes (tools/collect_profiles.sh) -> <dir>/prof
# 2. the counters' summary becomes profiles/rNN_pmc_traffic.json ON THE BOX, so that the bench lines below carry roofline.traffic
# 3. the default bench line, twice (plain `python bench.py` = the driver's command), and the two-rank rehearsals on one card
set -o pipefail
OUT=${1:-gpurun_out/evidence}
R=${2:-r04}
mkdir -p "$OUT"
bash tools/collect_profiles.sh "$OUT/prof" > "$OUT/collect.log" 2>&1 || { echo "collect_profiles failed"; tail -5 "$OUT/collect.log"; exit 1; }
cp "$OUT/prof/pmc_traffic.json" "profiles/${R}_pmc_traffic.json"
python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || { echo "bench failed"; tail -5 "$OUT/bench_n1.err"; exit 1; }
python bench.py --steps 5 --warmup 2 --no-2p24 --no-g2 --no-cpu-baseline > "$OUT/bench_n1_steps5.json" 2> "$OUT/bench_n1_steps5.err" || exit 1
GH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 1 --log-n 18 --no-ntt > "$OUT/bench_n2_gloo.json" 2> "$OUT/bench_n2_gloo.err" || { echo "n2 gloo failed"; tail -5 "$OUT/bench_n2_gloo.err"; }
python bench.py --gpus 2 --steps 4 --warmup 1 --log-n 18 --no-ntt > "$OUT/bench_n2_rccl.json" 2> "$OUT/bench_n2_rccl.err" || { echo "n2 rccl failed"; tail -5 "$OUT/bench_n2_rccl.err"; }
python3 tools/f_rows_bench.py 20 > "$OUT/f_rows.txt" 2> "$OUT/f_rows.err" || { echo "f_rows failed"; tail -3 "$OUT/f_rows.err"; }
ls -la "$OUT"
