R=$(pwd); L=$R/ginger-lib_amd
cp $L/libginger_hip.so /tmp/final.so
for v in final split2 split1 final; do
  if [ $v = final ]; then cp /tmp/final.so $L/libginger_hip.so; else cp $L/libginger_hip_ab_$v.so $L/libginger_hip.so; fi
  echo "== $v"; python3 tools/g2_probe.py mnt4753_g2 20 3 2>&1 | tail -2 | cut -c1-140; python3 tools/g2_probe.py mnt6753_g2 19 2 2>&1 | tail -1 | cut -c1-140
done
cp /tmp/final.so $L/libginger_hip.so
