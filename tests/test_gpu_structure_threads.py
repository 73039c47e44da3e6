"""The host side of fgo_build (fgo_structure.cpp: pair records, block graph, edge slots, half-edge lists -- counted and placed
with atomic increments on all host threads, every list sorted afterwards) must hand the device the same tables whatever the
thread count: same structure statistics, bit-identical LM trajectory and estimate.  FGO_HOST_THREADS is read once per process,
hence the subprocesses."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
import graph_slam_amd as G
from tests.util import vio_graph
from tests.test_gpu_imu import vio_gpu
out = {}
# g2o semantics: a pose graph with repeated edges on some pairs (duplicate groups) and a hub vertex
g = G.synth_manhattan3d(6000, 5, 4, 11)
ei, ej = g["ei"].astype(np.int64), g["ej"].astype(np.int64)
rng = np.random.default_rng(3)
dup = rng.choice(len(ei), 400, replace=False)
hub_j = rng.choice(np.arange(10, 6000), 1500, replace=False)
ei = np.concatenate([ei, ei[dup], np.full(len(hub_j), 5)]); ej = np.concatenate([ej, ej[dup], hub_j])
meas = np.concatenate([g["meas"], g["meas"][dup], np.tile(g["meas"][:1], (len(hub_j), 1))])
info = np.concatenate([g["info"], g["info"][dup], np.tile(g["info"][:1] * 1e-4, (len(hub_j), 1))])
fixed = np.zeros(6000, np.uint8); fixed[0] = 1
gr = G.Graph(); gr.add_poses(g["poses"], fixed); gr.add_edges(ei, ej, meas, info)
for _ in range(3): gr.optimize(2)
st = gr.stats()
out["g2o"] = [st.nnz_L_blocks, st.n_update_ops, st.n_levels, st.n_tasks, hashlib.sha256(gr.get_poses().tobytes()).hexdigest(), [float(x) for x in gr.trace()[0]]]
# GTSAM semantics: keyframes + velocities + biases + IMU factors + planes
v = vio_graph(np.random.default_rng(7), n_kf=120, with_planes=True)
gv = vio_gpu(v)
rc, st2 = gv.optimize_gtsam(4)
s0 = gv.stats()
ids = np.arange(len(v["values"]))
out["vio"] = [s0.nnz_L_blocks, s0.n_update_ops, s0.n_levels, rc, st2.trials, repr(gv.error()), hashlib.sha256(gv.get_poses(ids=ids).tobytes()).hexdigest()]
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(threads):
    env = dict(os.environ, FGO_HOST_THREADS=str(threads))
    p = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_device_tables_do_not_depend_on_the_host_thread_count():
    ref = _run(1)
    for threads in (3, 16):
        assert _run(threads) == ref


ORDER_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
import graph_slam_amd as G
g = G.synth_manhattan3d(60000, 5, 4, 17)
fixed = np.zeros(60000, np.uint8); fixed[0] = 1
gr = G.Graph(); gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"].astype(np.int64), g["ej"].astype(np.int64), g["meas"], g["info"])
for _ in range(2): gr.optimize(2)
st = gr.stats()
print("RESULT " + json.dumps([st.nnz_L_blocks, st.n_update_ops, st.n_levels, hashlib.sha256(gr.get_poses().tobytes()).hexdigest(), [float(x) for x in gr.trace()[0]]]))
""" % ROOT


def test_factor_does_not_depend_on_the_launch_order_of_the_triangle_kernels():
    """PanelPlan::tri_order (fgo_structure.cpp): the throughput triangle kernels of a wide level take its panels in width order
    (narrowest first for one wave per panel, widest first for eight waves) -- a permutation of independent workgroups, so the
    factor, the LM trajectory and the estimate must be the same bit for bit as in task order (FGO_TUNE tri_lpt=0), and with the
    orders forced the other way round (2 / 3).  60 000 poses: levels of > 3 x 256 panels (k_panel_tri1) and of > 256 (k_panel_tri<8>)."""
    res = []
    # (leaf_lpt: the light sub-trees of a leaf level by descending work, PanelPlan::leaf_lpt -- again only the order of independent workgroups)
    for tune in ("tri_lpt=0,leaf_lpt=0", "tri_lpt=1", "tri_lpt=2,leaf_lpt=0", "tri_lpt=3"):
        env = dict(os.environ, FGO_TUNE=tune)
        p = subprocess.run([sys.executable, "-c", ORDER_SCRIPT], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        res.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]))
    assert res[0][2] >= 10 and res[0][0] > 1000000        # (a structure with wide levels)
    for r in res[1:]:
        assert r == res[0]
