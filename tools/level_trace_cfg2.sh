#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lvl; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/t -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-iters 0 --phase-reps 1 --repeats 1 --extras 0 > /dev/null 2>&1
cd $ROOT; python tools/level_breakdown2.py $(ls -t $OUT/t/*/*.db | head -1) | tail -34
