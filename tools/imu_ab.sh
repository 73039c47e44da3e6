#!/bin/bash
# developer A/B: kernel times of the IMU linearisation with alternative libraries (FGO_LIB), cfg 4 graph; $@ = libraries
ROOT=$(pwd); OUT=$ROOT/gpurun_out/imu_ab; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  FGO_LIB=$ROOT/$lib timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/$tag -- python $ROOT/tools/run_scenarios.py vio --kf 50000 --iters 2 > $OUT/$tag.log 2>&1
  python $ROOT/tools/rocpd_summary.py "$(ls -t $OUT/$tag/*/*.db | head -1)" | grep -E "k_imu|k_zero\(" | cut -c1-150 | sed "s/^/$tag: /"
  rm -rf $OUT/$tag
done
