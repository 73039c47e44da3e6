"""Per-rank device time of the distributed mode measured on ONE GPU: rank r of `world` runs alone with a no-op transport
(the numbers it computes are meaningless -- the sums over the ranks are missing -- but it launches exactly the kernels rank
r launches on its own GPU, so its device time per LM trial is what that GPU would spend, collectives excluded).
usage: dist_rank_timing.py <poses> <world> [ranks...]   prints one JSON line per rank"""
import json, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, ".")
import graph_slam_amd as G

n = int(sys.argv[1]); world = int(sys.argv[2])
ranks = [int(x) for x in sys.argv[3:]] or list(range(world))
g = G.synth_manhattan3d(n, 5, 4, seed=42 if n != 1000000 else 45)
fixed = np.zeros(n, np.uint8); fixed[0] = 1
for r in ranks:
    gr = G.Graph()
    nbytes = [0]
    def hook(ptr, cnt):
        nbytes[0] += 8 * cnt
        return 0
    if world > 1:
        gr.set_shard(r, world, hook)
    gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    gr.chi2()
    sst = gr.stats()
    gr.optimize(1)
    nbytes[0] = 0
    t0 = time.perf_counter()
    rc, st = gr.optimize(3)
    dt = time.perf_counter() - t0
    print(json.dumps({"poses": n, "world": world, "rank": r, "trials": st.trials, "device_ms_per_trial": st.reserved[0] / max(st.trials, 1),
                      "domain_phase_ms_per_trial": st.ms_factor / max(st.trials, 1), "top_phase_incl_backward_update_linearise_ms_per_trial": st.ms_solve / max(st.trials, 1),
                      "wall_ms_per_trial": 1e3 * dt / max(st.trials, 1), "collective_bytes_per_trial": nbytes[0] / max(st.trials, 1),
                      "levels": sst.n_levels, "t_symbolic": sst.t_symbolic}))
    gr.close()
