import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import graph_slam_amd as G
n = 100000
g = G.synth_manhattan3d(n, 5, 4, 42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
for rep in range(3):
    t0 = time.time()
    gr = G.Graph(); gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    t1 = time.time()
    st = gr.optimize(2)[1]
    t2 = time.time()
    for i in range(9): gr.optimize(2)
    t3 = time.time()
    print("add %.1f ms, first optimize(2) %.1f ms (t_symbolic %.1f, t_upload %.1f), 9 more %.1f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*st.t_symbolic, 1e3*st.t_upload, 1e3*(t3-t2)), file=sys.stderr)
