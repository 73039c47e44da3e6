#!/usr/bin/env python3
"""Developer tool: run BASELINE configs 3 (BA) / 4 (VIO) at a given scale on the GPU and print timings."""
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
ap.add_argument("which", choices=["ba", "vio"])
ap.add_argument("--kf", type=int, default=1000)
ap.add_argument("--pts", type=int, default=50000)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
t0 = time.time()
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
print(json.dumps(out))
