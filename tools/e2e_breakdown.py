"""Where the wall-clock of a fresh context goes: add, first optimize (structure + upload + graph capture), later calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graph_slam_amd as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = G.synth_manhattan3d(n, 5, 4, 42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
for rep in range(3):
    t0 = time.time()
    gr = G.Graph(); gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    t1 = time.time()
    calls = []
    for i in range(10):
        ta = time.time(); st = gr.optimize(2)[1]; calls.append((1e3 * (time.time() - ta), st.trials, 1e3 * st.t_symbolic, 1e3 * st.t_upload, st.reserved[0]))
    print("add %.1f ms; optimize(2) calls (wall ms, trials, t_symbolic, t_upload, device ms): %s" %
          (1e3 * (t1 - t0), " | ".join("%.1f %d %.1f %.1f %.1f" % c for c in calls)), file=sys.stderr)
    print("total %.1f ms -> %.1f it/s end to end" % (1e3 * (time.time() - t0), 20 / (time.time() - t0)), file=sys.stderr)
    gr.close()
