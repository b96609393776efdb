OUT=gpurun_out/r03_shard_final.txt
: > $OUT
run() { python3 tools/acc_probe.py $1 $2 1 ${3:-6} 2 nocheck 2>&1 | grep -E "single|batch" | tail -2 >> $OUT; }
run mnt6753_g1 19 8; run mnt6753_g1 20 8; run mnt6753_g1 21 6; run mnt6753_g1 22 6
run mnt6753_g2 19 4; run mnt6753_g2 20 4; run mnt6753_g2 21 3; run mnt6753_g2 22 3
cat $OUT
