"""debug: distributed mode as threads on one GPU vs the single-GPU run (prints LM statistics)"""
import sys, threading
import numpy as np
import torch
sys.path.insert(0, ".")
import graph_slam_amd as G
from tests.test_gpu_parity import synth, make_gpu

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
g = synth(n, 5, 4, seed=12)
ref = make_gpu(g)
for it in range(4):
    rc, st = ref.optimize(2)
    print("ref", rc, st.chi2_initial, st.chi2_final, st.trials, st.lambda_final, st.terminated, ref.trace())
staging = [None] * world
barrier = threading.Barrier(world)
out = [None] * world
def run(rank):
    gr = make_gpu(g)
    def hook(ptr, nn):
        t = G.device_tensor(ptr, nn)
        staging[rank] = t.cpu()
        barrier.wait()
        total = staging[0].clone()
        for q in range(1, world): total += staging[q]
        barrier.wait()
        t.copy_(total)
        return 0
    gr.set_shard(rank, world, hook)
    out[rank] = []
    for it in range(4):
        rc, st = gr.optimize(2)
        out[rank].append((rc, st.chi2_initial, st.chi2_final, st.trials, st.lambda_final, st.terminated, gr.trace()))
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
for r in range(world):
    for o in out[r]: print("rank", r, o)
