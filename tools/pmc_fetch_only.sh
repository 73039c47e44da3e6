#!/bin/bash
# FETCH_SIZE pass only (HBM-side reads per kernel launch); prints the accumulate kernels
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcf; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-iters 0 --phase-reps 1 > /dev/null 2>&1
cd $ROOT
DB=$(ls -t $OUT/f/*/*.db | head -1)
python tools/pmc_traffic.py $DB $DB | grep -E "kernel|k_chol_acc|k_panel|k_linearize|sweep"
rm -rf $OUT/f
