#!/bin/bash
# rocprofv3 kernel tables of BASELINE configs 3 (bundle adjustment) and 4 (visual-inertial + planes); outputs in gpurun_out/prof_sc/
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_sc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/ba -- python $ROOT/tools/run_scenarios.py ba --kf 10000 --pts 500000 --iters 5 > $OUT/ba.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/vio -- python $ROOT/tools/run_scenarios.py vio --kf 50000 --iters 5 > $OUT/vio.log 2>&1
cd $ROOT
for w in ba vio; do python tools/rocpd_summary.py "$(ls -t $OUT/$w/*/*.db | head -1)" > $OUT/${w}_kernel_stats.txt; grep '^{' $OUT/$w.log | tail -1 > $OUT/$w.json; rm -rf $OUT/$w; done
head -25 $OUT/ba_kernel_stats.txt
