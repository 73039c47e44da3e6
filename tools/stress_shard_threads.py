#!/usr/bin/env python3
"""Developer tool: two shard ranks as host threads on one GPU, repeated; reports any divergence between the ranks."""
import os, sys, threading, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("FGO_PKG_ROOT"): sys.path.insert(0, os.environ["FGO_PKG_ROOT"])   # an older build of the package, for bisecting
import numpy as np, torch
import graph_slam_amd as G
from tests.test_gpu_parity import make_gpu
from tests.test_gpu_shard import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = synth(2000, 5, 4, seed=12)
bad = 0
for rep in range(reps):
    world = 2
    staging = [None] * world
    barrier = threading.Barrier(world)
    out = [None] * world
    calls = [0, 0]
    log = [[], []]

    def run(rank):
        try:
            gr = make_gpu(g)
            def hook(ptr, n):
                calls[rank] += 1
                t = G.device_tensor(ptr, n)
                staging[rank] = t.cpu()
                barrier.wait(timeout=10)
                total = staging[0] + staging[1]
                barrier.wait(timeout=10)
                t.copy_(total)
                if n == 1: log[rank].append(float(total[0]))
                return 0
            gr.set_shard(rank, world, hook)
            rc, st = gr.optimize(4)
            out[rank] = (rc, st.trials, np.array(gr.trace()[0]), np.array(gr.trace()[1]))
        except Exception as e:
            out[rank] = ("EXC", repr(e))
            barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    ok = out[0] is not None and out[1] is not None and out[0][0] != "EXC" and out[1][0] != "EXC" and calls[0] == calls[1]
    if not ok:
        bad += 1
        print("rep", rep, "calls", calls, "out0", out[0], "out1", out[1])
        print(" chi2 log rank0", log[0]); print(" chi2 log rank1", log[1])
print("anomalies: %d / %d" % (bad, reps))
