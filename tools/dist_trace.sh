#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/dtrace; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
PYTHONPATH=$ROOT timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t -- python $ROOT/tools/dist_rank_timing.py 100000 8 0 > /dev/null 2>&1
cd $ROOT; python tools/rocpd_summary.py $(ls -t $OUT/t/*/*.db | head -1) | head -32 | cut -c1-150
