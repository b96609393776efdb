# Timing ablations of the G2 backward round kernel (results are wrong on purpose): per variant the kernel stats of one MSM.
# Libraries libginger_hip_ab_<variant>.so are built with GH_ASM_ABLATE=<variant> (asmgen/g2_rounds.py).  bash tools/ab_ablate.sh OUT
OUT=${1:-gpurun_out/ab_ablate}
R=$(pwd); L=$R/ginger-lib_amd
mkdir -p $OUT
cp $L/libginger_hip.so /tmp/final.so
cd /tmp && export TMPDIR=/tmp
for v in final parks loads stores bperm; do
  if [ $v = final ]; then cp /tmp/final.so $L/libginger_hip.so; else cp $L/libginger_hip_ab_$v.so $L/libginger_hip.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$v -o run -- python3 $R/tools/g2_probe.py mnt4753_g2 20 2 > $R/$OUT/$v.log 2>&1
  echo "== $v"; grep "gh_asm_aff_f2" $R/$OUT/$v/run_kernel_stats.csv | cut -d, -f1-4
done
cp /tmp/final.so $L/libginger_hip.so
