#!/bin/bash
# HBM traffic of the factor sweep (two rocprofv3 PMC passes) + it/s for an environment setting: tools/pmc_ab.sh TAG VAR=VALUE ...
ROOT=$(pwd); tag=$1; shift; OUT=$ROOT/gpurun_out/pmc_ab/$tag; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
env "$@" timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-iters 0 --phase-reps 1 --repeats 1 > /dev/null 2>&1
env "$@" timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-iters 0 --phase-reps 1 --repeats 1 > /dev/null 2>&1
cd $ROOT
python tools/pmc_traffic.py "$(ls -t $OUT/f/*/*.db | head -1)" "$(ls -t $OUT/w/*/*.db | head -1)" > $OUT/traffic.txt
echo "$tag: $(tail -1 $OUT/traffic.txt)  $(env "$@" python bench.py --cpu-iters 0 --repeats 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f it/s, sweep %.3f ms' % (d['value'], d['roofline']['ms_per_pass']))")"
rm -rf $OUT/f $OUT/w
