for kv in "X=0" "GH_AFF_LEFTOVER=1.0" "GH_AFF_LEFTOVER=2.0" "GH_AFF_LEFTOVER=3.5" "GH_AFF_BMIN=4" "GH_AFF_BMIN=16" "GH_AFF_FINISH_MAX=32" "GH_AFF_FINISH_MAX=128"; do
  echo "== $kv"
  env $kv timeout -k 10 200 python3 tools/g2_probe.py mnt6753_g2 19 1 0 4 2>&1 | tail -1
  env $kv timeout -k 10 200 python3 tools/g2_probe.py mnt4753_g2 20 1 0 4 2>&1 | tail -1
done
