#!/bin/bash
# per-kernel totals of the headline bench for two builds of libfgo (FGO_LIB): tools/ab_kernel_stats.sh <libA> <libB>
ROOT=$(pwd); OUT=$ROOT/gpurun_out/abk; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for tag in A B; do
  lib=$1; [ $tag = B ] && lib=$2
  FGO_LIB=$ROOT/$lib timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/$tag -- python $ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --cpu-iters 0 > /dev/null 2>&1
  python $ROOT/tools/rocpd_summary.py "$(ls -t $OUT/$tag/*/*.db | head -1)" > $OUT/$tag.txt
  rm -rf $OUT/$tag
done
paste <(cut -c1-60,104-140 $OUT/A.txt | head -18) <(cut -c104-140 $OUT/B.txt | head -18)
