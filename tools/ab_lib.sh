set -e
L=ginger-lib_amd
cp $L/libginger_hip.so /tmp/new.so
for round in 1 2; do
  cp $L/libginger_hip_base.so $L/libginger_hip.so; echo BASE; python3 tools/acc_probe.py mnt4753_g1 20 1 10 3 nocheck 2>&1 | grep -v precompute | tail -2 | cut -c1-150
  cp /tmp/new.so $L/libginger_hip.so; echo NEW; python3 tools/acc_probe.py mnt4753_g1 20 1 10 3 nocheck 2>&1 | grep -v precompute | tail -2 | cut -c1-150
done
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm_vs_oracle or golden or skewed or window_sizes" 2>&1 | tail -2
