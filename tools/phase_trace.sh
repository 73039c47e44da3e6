#!/bin/bash
# per-level kernel durations of the factor phase alone (fgo_bench_phase(1)): $1 = tag; env passes through
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import graph_slam_amd as G
n = 100000
g = G.synth_manhattan3d(n, 5, 4, 42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
gr = G.Graph(); gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
print(gr.bench_phase(1, 8))
PY
FGO_GRAPH=0 timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -- python $OUT/run.py > $OUT/log.txt 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/level_breakdown2.py $DB > $OUT/levels.txt 2>&1
rm -rf $OUT/trace
tail -2 $OUT/log.txt | head -1
