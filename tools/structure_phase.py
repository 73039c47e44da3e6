#!/usr/bin/env python3
"""Developer tool: time of the host structure phase (ordering + symbolic + upload) on the benchmark graph.
FGO_SYM_PROFILE=1 prints the sections of the symbolic phase, FGO_HOST_THREADS sets the host thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_slam_amd as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = G.synth_manhattan3d(n, 5, 4, 42)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
for rep in range(2):
    gr = G.Graph(verbose=1)
    gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    t = time.time(); c = gr.chi2(); st = gr.stats()
    print("poses %d: first chi2 (build + upload + one kernel) %.3f s; t_symbolic %.3f t_upload %.3f" % (n, time.time() - t, st.t_symbolic, st.t_upload))
    gr.close()
