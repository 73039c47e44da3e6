"""Where a grow-by-10 step of the g2o cadence goes (bench.py end_to_end_growing_graph): hand-over, in-place extension, first and later
optimize(2) calls, device time per trial with and without the growth reserve."""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graph_slam_amd as G
from bench import pose_compose

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = G.synth_manhattan3d(n, 5, 4, seed=42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
n_steps, step = 5, 10
n0 = n - n_steps * step
old = g["ej"] < n0
local = (g["ej"] - g["ei"]) <= 10
for reserve in (0, 384):
    gr = G.Graph()
    gr.set_growth(reserve, 64)
    gr.add_poses(g["poses"][:n0], fixed[:n0])
    gr.add_edges(g["ei"][old], g["ej"][old], g["meas"][old], g["info"][old])
    t0 = time.perf_counter(); rc, st = gr.optimize(2); t1 = time.perf_counter()
    print("reserve %d: first optimize(2) %.1f ms (symbolic %.1f upload %.1f), levels %d, nnzL %d, ops %d" % (reserve, 1e3 * (t1 - t0), 1e3 * st.t_symbolic, 1e3 * st.t_upload, st.n_levels, st.nnz_L_blocks, st.n_update_ops))
    for _ in range(9):
        rc, st = gr.optimize(2)
    tt = []
    for _ in range(5):
        t0 = time.perf_counter(); rc, st = gr.optimize(2); tt.append(1e3 * (time.perf_counter() - t0))
    print("  steady optimize(2): %s ms wall, %d trials, %.3f ms device per trial" % (" ".join("%.1f" % t for t in tt), st.trials, st.reserved[0] / max(st.trials, 1)))
    if reserve == 0:
        gr.close(); continue
    odo = {int(b): k for k, (a, b) in enumerate(zip(g["ei"], g["ej"])) if b - a == 1 and b >= n0}
    for sidx in range(n_steps):
        lo, hi = n0 + sidx * step, n0 + (sidx + 1) * step
        t0 = time.perf_counter()
        prev = gr.get_poses(ids=np.array([lo - 1]))[0]
        t1 = time.perf_counter()
        new = []
        for v in range(lo, hi):
            prev = pose_compose(prev, g["meas"][odo[v]]); new.append(prev)
        em = (g["ej"] >= lo) & (g["ej"] < hi) & local
        t2 = time.perf_counter()
        gr.add_poses(np.array(new), np.zeros(step, np.uint8), ids=np.arange(lo, hi))
        gr.add_edges(g["ei"][em], g["ej"][em], g["meas"][em], g["info"][em])
        t3 = time.perf_counter()
        calls = []
        for k in range(10):
            ta = time.perf_counter(); rc, st = gr.optimize(2); calls.append(1e3 * (time.perf_counter() - ta))
            if k == 0: st0 = st
        print("  step %d: get pose %.1f, numpy %.1f, add %.1f ms; optimize(2) calls %s ms; extension %.2f ms host, rebuilt %d, trials of call 0: %d, %.3f ms device per trial"
              % (sidx, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), " ".join("%.1f" % c for c in calls), 1e3 * st0.t_symbolic, st0.structure_rebuilt, st0.trials, st0.reserved[0] / max(st0.trials, 1)))
    gr.close()
