#!/usr/bin/env python3
"""Developer tool: run BASELINE configs 3 (BA) / 4 (VIO) at a given scale on the GPU and print timings; `isam` replays a
pose graph record by record (new poses + their factors, then fgo_isam2_update, as the reference's drivers do)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import graph_slam_amd as G
from graph_slam_amd import scenarios as S

ap = argparse.ArgumentParser()
ap.add_argument("which", choices=["ba", "vio", "isam", "hubs", "torus"])
ap.add_argument("--per-update", type=int, default=1, help="isam: new poses per update")
ap.add_argument("--kf", type=int, default=1000)
ap.add_argument("--pts", type=int, default=50000)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
t0 = time.time()
if a.which == "isam":
    g = G.synth_manhattan3d(a.kf, 5, 4, 42)
    W = np.diag([1 / 0.01 ** 2] * 3 + [1 / 0.02 ** 2] * 3)
    iu = np.triu_indices(6)
    info = W[iu]
    ei, ej = g["ei"].astype(np.int64), g["ej"].astype(np.int64)
    newest = np.maximum(ei, ej)
    order = np.argsort(newest, kind="stable")
    ei, ej, meas, newest = ei[order], ej[order], g["meas"][order], newest[order]
    start = np.searchsorted(newest, np.arange(a.kf + 1))
    gr = G.Graph()
    gr.add_poses(g["poses"][:1])
    gr.add_prior(0, g["poses"][0], np.diag([1e6] * 6)[iu])
    wall, sym, dev, relin = [], [], [], []
    have = 1
    while have < a.kf:
        k = min(a.kf, have + a.per_update)
        # initial value of a new pose: odometry chained onto the current estimate of its predecessor (addToGTSAM)
        gr.add_poses(g["poses"][have:k], ids=np.arange(have, k))
        e0, e1 = start[have], start[k]
        gr.add_edges(ei[e0:e1], ej[e0:e1], meas[e0:e1], np.tile(info, (e1 - e0, 1)), tangent_order=G.FGO_TANGENT_GTSAM)
        have = k
        t = time.time()
        st = gr.isam2_update(0.1)
        wall.append(time.time() - t); sym.append(st.t_symbolic + st.t_upload); dev.append(st.reserved[0]); relin.append(st.reserved[1])
    wall, sym, dev = np.array(wall), np.array(sym), np.array(dev)
    q = len(wall) // 4
    t = time.time(); st = gr.isam2_update(0.1); t_static = time.time() - t
    print(json.dumps({"which": "isam", "poses": a.kf, "edges": int(len(ei)), "updates": len(wall), "per_update": a.per_update,
                      "wall_ms_mean": 1e3 * wall.mean(), "wall_ms_last_quarter": 1e3 * wall[-q:].mean(),
                      "structure_ms_last_quarter": 1e3 * sym[-q:].mean(), "device_ms_last_quarter": float(dev[-q:].mean()),
                      "relinearised_per_update_mean": float(np.mean(relin)), "update_without_new_factors_ms": 1e3 * t_static,
                      "error": gr.error(), "total_s": time.time() - t0}))
    sys.exit(0)
if a.which in ("hubs", "torus"):
    # g2o-semantics pose graphs of other shapes (tests/test_gpu_scenarios.py::test_other_topologies): --kf = poses (hubs) or
    # the side of the torus grid
    g = S.hub_graph(n=a.kf) if a.which == "hubs" else S.torus_graph(nu=a.kf, nv=a.kf)
    n = len(g["poses"])
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    gr = G.Graph()
    gr.add_poses(g["poses"], fixed)
    gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    c0 = gr.chi2()
    st0 = gr.stats()
    for _ in range(a.iters):
        rc, st = gr.optimize(2)
    out = {"which": a.which, "poses": n, "edges": int(len(g["ei"])), "t_symbolic": st0.t_symbolic, "chi2_0": c0, "chi2": st.chi2_final,
           "nnz_L": st0.nnz_L_blocks, "ops": st0.n_update_ops, "levels": st0.n_levels, "tasks": st0.n_tasks}
    out["phases_ms"] = {nm: gr.bench_phase(k, 2) for k, nm in ((0, "linearize"), (1, "factor"), (2, "solve"))}
    print(json.dumps(out))
    sys.exit(0)
if a.which == "ba":
    p = S.ba_problem(a.kf, a.pts)
    t1 = time.time()
    gr = S.ba_graph(p)
    extra = {"observations": int(len(p["obs_uv"]))}
else:
    p = S.vio_problem(a.kf)
    t1 = time.time()
    gr, nobs = S.vio_graph(p)
    extra = {"plane_factors": nobs}
t2 = time.time()
e0 = gr.error()
t3 = time.time()
st0 = gr.stats()
rc, st = gr.optimize_gtsam(a.iters)
t4 = time.time()
out = {"which": a.which, "kf": a.kf, "gen_s": t1 - t0, "assemble_s": t2 - t1, "build+error_s": t3 - t2, "t_symbolic": st0.t_symbolic,
       "t_upload": st0.t_upload, "iters": rc, "trials": st.trials, "opt_s": t4 - t3, "error0": e0, "error": gr.error(),
       "n_free": st0.n_free, "nnz_H": st0.nnz_H_blocks, "nnz_L": st0.nnz_L_blocks, "ops": st0.n_update_ops, "levels": st0.n_levels,
       "tasks": st0.n_tasks}
out.update(extra)
out["phases_ms"] = {n: gr.bench_phase(k, 2) for k, n in ((0, "linearize"), (1, "factor"), (2, "solve"))}
# roofline of every phase and of the dominant one (configs 3 / 4 had phase times only): algorithmic bytes by SURVEY 8d's
# accounting as the structure phase sums them (fgo_stats.bytes_*: factor records and values read once, H / L / W written once, L re-read
# once by the fused forward solve; BA: W 144 B per observation three times in the reduction, once in the back-substitution) over the
# phase time measured with HIP events on the library's stream (fgo_bench_phase)
HBM_PEAK_GBS = 8000.0
byt = {"linearize": st0.bytes_linearize, "factor": st0.bytes_factor, "solve": st0.bytes_solve}
gbs = {n: byt[n] / (out["phases_ms"][n] * 1e-3) / 1e9 for n in byt}
dom = max(out["phases_ms"], key=lambda n: out["phases_ms"][n])
out["roofline"] = {"bound": "hbm", "kernel": dom + " phase", "achieved": gbs[dom], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs[dom] / HBM_PEAK_GBS,
                   "algorithmic_bytes_per_pass": byt[dom], "ms_per_pass": out["phases_ms"][dom], "phases_GBs": gbs,
                   "trial_GBs": sum(byt.values()) / (sum(out["phases_ms"].values()) * 1e-3) / 1e9, "traffic": None}
print(json.dumps(out))
