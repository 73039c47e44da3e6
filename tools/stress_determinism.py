#!/usr/bin/env python3
"""Developer tool: the same graph optimised in fresh contexts must give bit-identical traces (no FP atomics, fixed
summation orders).  Prints the distinct (chi2 trace, trials) outcomes seen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_slam_amd as G
from tests.test_gpu_parity import make_gpu
from tests.test_gpu_shard import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
g = synth(n, 5, 4, seed=12)
seen = {}
keep = []
for rep in range(reps):
    gr = make_gpu(g)
    rc, st = gr.optimize(4)
    key = (tuple(gr.trace()[0]), st.trials)
    seen[key] = seen.get(key, 0) + 1
    keep.append(torch.empty(int(np.random.randint(1, 64)) << 20, dtype=torch.uint8, device="cuda").fill_(0xFF))   # dirty, shifting heap
    if len(keep) > 4: keep.pop(0)
for k, v in seen.items(): print(v, "x", k)
print("distinct outcomes:", len(seen))
