#!/bin/bash
# HBM traffic per kernel (two rocprofv3 PMC passes each: FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of BASELINE configs 3
# (bundle adjustment) and 4 (visual-inertial + planes); outputs in gpurun_out/pmc_sc/
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_sc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
run() {  # tag, scenario arguments
  tag=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/${tag}_$ctr -- python $ROOT/tools/run_scenarios.py "$@" > $OUT/${tag}_$ctr.log 2>&1
  done
  python $ROOT/tools/pmc_traffic.py "$(ls -t $OUT/${tag}_FETCH_SIZE/*/*.db | head -1)" "$(ls -t $OUT/${tag}_WRITE_SIZE/*/*.db | head -1)" > $OUT/${tag}_pmc_traffic.txt
  rm -rf $OUT/${tag}_FETCH_SIZE $OUT/${tag}_WRITE_SIZE
}
run cfg3_ba ba --kf 10000 --pts 500000 --iters 3
run cfg4_vio vio --kf 50000 --iters 3
head -16 $OUT/cfg3_ba_pmc_traffic.txt; head -16 $OUT/cfg4_vio_pmc_traffic.txt
