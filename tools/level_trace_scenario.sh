#!/bin/bash
# per-level kernel durations of one factor sweep of a scenario (tools/run_scenarios.py): $1 = tag, rest = scenario arguments
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; rm -rf $OUT; mkdir -p $OUT
FGO_GRAPH=0 timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -- python $GRAFT_REPO_ROOT/tools/run_scenarios.py "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/level_breakdown2.py $DB > $OUT/levels.txt 2>&1
rm -rf $OUT/trace
