# One card at the shard sizes of BASELINE config 4 (2^22 pairs over 1 / 2 / 4 / 8 GPUs), chain keys, shift tables, one MSM and a
# pipelined batch:  bash tools/shard_final.sh [OUT]      (the predicted scaling curve of DESIGN.md section 6)
OUT=${1:-gpurun_out/r04_shard_final.txt}
: > $OUT
run() { python3 tools/g2_probe.py $1 $2 2 0 ${3:-6} 2>&1 | tail -2 >> $OUT; }
run mnt6753_g1 19 8; run mnt6753_g1 20 8; run mnt6753_g1 21 6; run mnt6753_g1 22 6
run mnt6753_g2 19 4; run mnt6753_g2 20 4; run mnt6753_g2 21 3; run mnt6753_g2 22 3
cat $OUT
