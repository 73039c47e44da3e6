import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import graph_slam_amd as G
sizes = [int(a) for a in sys.argv[1:]] or [2000, 20000]
for n in sizes:
    g = G.synth_manhattan3d(n + 50, 5, 4, 42)
    iu = np.triu_indices(6); info = np.diag([1e4] * 3 + [2500.] * 3)[iu]
    ei, ej = g["ei"].astype(np.int64), g["ej"].astype(np.int64); newest = np.maximum(ei, ej)
    gr = G.Graph(verbose=1)
    import os
    if os.environ.get("FGO_WILDFIRE"): gr.isam2_set_wildfire(float(os.environ["FGO_WILDFIRE"]))
    gr.add_poses(g["poses"][:n]); gr.add_prior(0, g["poses"][0], np.diag([1e6] * 6)[iu])
    m = newest < n
    gr.add_edges(ei[m], ej[m], g["meas"][m], np.tile(info, (m.sum(), 1)), tangent_order=G.FGO_TANGENT_GTSAM)
    gr.isam2_update(0.1)
    for it in range(15):                      # settle: until no variable is relinearised any more (the drivers' steady state)
        if gr.isam2_update(0.1).reserved[1] == 0: break
    for k in range(n, n + 8):
        gr.add_poses(g["poses"][k:k + 1], ids=[k]); m = newest == k
        gr.add_edges(ei[m], ej[m], g["meas"][m], np.tile(info, (m.sum(), 1)), tangent_order=G.FGO_TANGENT_GTSAM)
        t = time.time(); st = gr.isam2_update(0.1); print(n, "update wall %.2f ms  host structure/extension %.2f upload %.2f device %.2f (relin+linearise %.2f factor %.2f backward %.2f estimate+chi2 %.2f) rebuilt %d tasks re-run %d of %d" % (1e3 * (time.time() - t), 1e3 * st.t_symbolic, 1e3 * st.t_upload, st.reserved[0], st.ms_linearize, st.ms_factor, st.ms_solve, st.ms_update, st.structure_rebuilt, int(st.reserved[3]), st.n_tasks))
