#!/bin/bash
# per-level kernel durations of one factor sweep (rocprofv3 kernel trace + tools/level_breakdown2.py); $1 = output tag, env passes through
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-4} --warmup 1 --cpu-iters 0 --phase-reps 1 --repeats 1 ${BENCH_ARGS:-} > $OUT/bench.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/level_breakdown2.py $DB > $OUT/levels.txt 2>&1
rm -rf $OUT/trace
