"""The drop-in boundary, proven with the reference's OWN files (BASELINE.json: "test_g2o_graph /
test_ba_imu_graph link unchanged").

graph_slam_amd/host holds no copy or re-typing of the reference's wrappers any more: g2o/g2o_graph.cpp,
gtsam/gtsam_graph.cpp, gtsam/imu_base.cpp, gtsam/imu_vn100.cpp, gtsam/imu_MEMS.cpp and the three drivers
g2o/test_g2o_graph.cpp, gtsam/test_vro_imu_graph.cpp, gtsam/test_ba_imu_graph.cpp are compiled IN PLACE from
/root/reference (when that tree exists: this container) against API slices of g2o / GTSAM that forward to libfgo.so
(host/shim/g2o, host/shim/gtsam_lite.h, host/fgo_g2o.cpp, host/gtsam_bridge.cpp) plus stand-ins for the un-vendored
front-end packages.  The built binaries travel to the GPU box like any other in-tree .so.

CPU: everything builds and exports the reference's class surface.
GPU: BASELINE config 1 through the reference's CGraphG2O and through its unmodified driver; the reference's VIO and
BA+IMU drivers replay synthetic VRO / IMU logs, ISAM2 per record, and are cross-checked against an independent numpy
rebuild of the same graph through the C-ABI.
"""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "graph_slam_amd", "host")
REF_DRIVER = "/root/reference/g2o/test_g2o_graph.cpp"


def _make(*targets):
    return subprocess.run(["make", "-s", "-C", HOST] + list(targets), capture_output=True, text=True)


def test_host_library_builds_and_exports_surface():
    r = _make()
    assert r.returncode == 0, r.stderr[-2000:]
    if not os.path.exists(REF_DRIVER) and not os.path.exists(os.path.join(HOST, "libg2o_graph.so")):
        pytest.skip("reference tree not present and no prebuilt wrapper library")
    out = subprocess.run(["nm", "-DC", os.path.join(HOST, "libg2o_graph.so")], capture_output=True, text=True).stdout
    for sym in ["CGraphG2O::CGraphG2O()", "CGraphG2O::createOptimizer()", "CGraphG2O::firstNode(CCameraNode*)",
                "CGraphG2O::addNode(CCameraNode*)", "CGraphG2O::fakeOdoNode(CCameraNode*)", "CGraphG2O::optimizeGraph()",
                "CGraphG2O::addToGraph(MatchingResult&, bool)", "CGraphG2O::isSmallTrafo(MatchingResult&)",
                "CGraphG2O::error()", "CGraphG2O::camnodeSize()", "CGraphG2O::writeG2O(", "CGraphG2O::writeTrajectory(",
                "CGraphG2O::setWorld2Original(double)", "CGraphG2O::headerPLY(", "CGraphG2O::trajectoryPLY(",
                "CG2OParams::Instance()"]:
        assert sym in out, "missing symbol " + sym


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="reference tree not present (GPU box)")
def test_reference_driver_compiles_and_links_unchanged():
    r = _make("ref_driver_check")
    assert r.returncode == 0, r.stderr[-3000:]
    for b in ("_ref_test_g2o_graph", "_ref_test_vro_imu_graph", "_ref_test_ba_imu_graph", "libg2o_graph.so", "libgtsam_graph.so"):
        assert os.path.exists(os.path.join(HOST, b)), b


def _oracle_schedule(n, lookback):
    import graph_slam_amd as G
    from tests import orc_binding as orc
    g = G.synth_manhattan3d(n, lookback, 0, seed=42)
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    po = orc.Problem(g["poses"], fixed, g["ei"].astype(np.int32), g["ej"].astype(np.int32), g["meas"], g["info"])
    before = po.chi2()
    for _ in range(10):                       # CGraphG2O::optimizeGraph: 10 x optimize(2)
        po.optimize(2)
    return before, po.chi2(), po.get_poses(), len(g["ei"])


@pytest.mark.gpu
def test_config1_through_cgraphg2o(tmp_path):
    assert _make().returncode == 0
    prefix = str(tmp_path / "cfg1")
    r = subprocess.run([os.path.join(HOST, "run_g2o_graph"), "1000", "4", "0", prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    before, after, poses, n_edges = _oracle_schedule(1000, 4)
    assert res["nodes"] == 1000 and res["keyframes"] == 1000 and res["fake"] == 0
    assert abs(res["chi2_before"] - before) <= 1e-9 * before      # wrapper rebuilt exactly the synthetic graph
    assert abs(res["chi2_after"] - after) <= 1e-8 * after         # north star: 1e-6
    # file writers: trajectory log "id x y z qx qy qz qw seq_id", .g2o, PLY
    traj = np.loadtxt(prefix + "_trajectory.log")
    assert traj.shape == (1000, 9)
    # the writer uses the stream's default 6 significant digits, like the reference (g2o_graph.cpp:302-303)
    np.testing.assert_allclose(traj[:, 1:4], poses[:, :3], rtol=1e-5, atol=1e-5)
    g2o_lines = open(prefix + ".g2o").read().splitlines()
    assert sum(l.startswith("VERTEX_SE3:QUAT") for l in g2o_lines) == 1000
    assert sum(l.startswith("EDGE_SE3:QUAT") for l in g2o_lines) == n_edges
    assert sum(l.startswith("FIX") for l in g2o_lines) == 1
    ply = open(prefix + "_after.ply").read().splitlines()
    assert ply[0] == "ply" and "element vertex 1000" in ply[2] and len(ply) == 10 + 1000


def _oracle_schedule_with_vo_failures(n, lookback, fail):
    """the graph CGraphG2O builds when the frames in `fail` match nothing older: their incoming edges are missing and
    fakeOdoNode (g2o/g2o_graph.cpp:136-157) ties each to its predecessor with an identity edge of information 1e-3 * I;
    estimates are chained through the odometry edges (identity across a failure)"""
    import graph_slam_amd as G
    from graph_slam_amd import scenarios as S
    from tests import orc_binding as orc
    g = G.synth_manhattan3d(n, lookback, 0, seed=42)
    ei, ej = g["ei"].astype(np.int64), g["ej"].astype(np.int64)
    assert (ej > ei).all()
    keep = ~np.isin(ej, list(fail))
    odo = dict(((int(a), int(b)), k) for k, (a, b) in enumerate(zip(ei, ej)) if b == a + 1)
    fei = np.array([f - 1 for f in fail]); fej = np.array(list(fail))
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    w = np.zeros(21); w[[0, 6, 11, 15, 18, 20]] = 1e-3
    E_i = np.concatenate([ei[keep], fei]); E_j = np.concatenate([ej[keep], fej])
    meas = np.concatenate([g["meas"][keep], np.tile(ident, (len(fail), 1))])
    info = np.concatenate([g["info"][keep], np.tile(w, (len(fail), 1))])
    poses = np.zeros((n, 7)); poses[0] = g["poses"][0]
    for k in range(1, n):
        z = ident if k in fail else g["meas"][odo[(k - 1, k)]]
        poses[k, :3] = poses[k - 1, :3] + S._quat_rot(poses[k - 1, 3:], z[:3])
        q = S._quat_mul(poses[k - 1, 3:], z[3:]); poses[k, 3:] = q / np.linalg.norm(q)
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    po = orc.Problem(poses, fixed, E_i.astype(np.int32), E_j.astype(np.int32), meas, info)
    before = po.chi2()
    for _ in range(10):
        po.optimize(2)
    return before, po.chi2(), po.get_poses(), len(E_i)


@pytest.mark.gpu
def test_vo_failure_goes_through_fake_odo_node(tmp_path):
    """VERDICT r2 missing #6: frames that match nothing older (featureless area) -> CGraphG2O::addNode returns FAIL_KF, the
    driver calls fakeOdoNode (g2o/test_g2o_graph.cpp:90-95), which adds an identity edge with information 1e-3 * I --
    seven orders of magnitude weaker than the VO edges (SURVEY.md §7 'hard parts': ill-conditioned edges).  The
    reference's own CGraphG2O on the GPU vs the oracle on the same graph."""
    assert _make().returncode == 0
    fail = [100, 101, 500, 777]                                   # incl. two consecutive failures
    env = dict(os.environ, FGO_SYNTH_VO_FAIL=",".join(map(str, fail)))
    prefix = str(tmp_path / "fake")
    r = subprocess.run([os.path.join(HOST, "run_g2o_graph"), "1000", "4", "0", prefix], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["fake"] == len(fail) and res["nodes"] == 1000 and res["keyframes"] == 1000 - len(fail)
    before, after, poses, n_edges = _oracle_schedule_with_vo_failures(1000, 4, fail)
    g2o_lines = open(prefix + ".g2o").read().splitlines()
    assert sum(l.startswith("EDGE_SE3:QUAT") for l in g2o_lines) == n_edges
    fake_lines = [l.split() for l in g2o_lines if l.startswith("EDGE_SE3:QUAT") and int(l.split()[2]) in fail and int(l.split()[1]) == int(l.split()[2]) - 1]
    assert len(fake_lines) == len(fail)
    for f in fake_lines:
        assert [float(x) for x in f[3:10]] == [0, 0, 0, 0, 0, 0, 1] and float(f[10]) == 1e-3
    assert abs(res["chi2_before"] - before) <= 1e-9 * before
    assert abs(res["chi2_after"] - after) <= 1e-7 * after          # north star: 1e-6
    traj = np.loadtxt(prefix + "_trajectory.log")
    np.testing.assert_allclose(traj[:, 1:4], poses[:, :3], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "_ref_test_g2o_graph")), reason="prebuilt reference driver not shipped")
def test_reference_driver_with_vo_failures(tmp_path):
    """the same injection through the reference's unmodified driver binary (its own FAIL_KF -> fakeOdoNode branch)"""
    fail = [100, 101, 500, 777]
    env = dict(os.environ, FGO_SYNTH_POSES="1000", FGO_SYNTH_LOOKBACK="4", sr_start_frame="1", sr_end_frame="1001",
               gt_lookback_nodes="4", gt_optimize_step="250", gt_output_dir=str(tmp_path), sr_data_name="fake",
               FGO_SYNTH_VO_FAIL=",".join(map(str, fail)))
    r = subprocess.run([os.path.join(HOST, "_ref_test_g2o_graph")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stderr.splitlines() if "optimization error is" in l]
    assert len(lines) == 2
    before = float(lines[0].split()[-1]); after = float(lines[1].split()[-1])
    assert after < before
    traj = np.loadtxt(str(tmp_path / "fake_vo_after_trajectory_g2o.log"))
    assert traj.shape == (1000, 9)                                 # the failed frames are in the graph (fake odometry)


@pytest.mark.gpu
def test_online_schedule_with_periodic_optimisation():
    """optimizeGraph every 100 keyframes, as the online driver does (test_g2o_graph.cpp:78-84): the structure is
    rebuilt each time new vertices arrive; the end result must be a converged graph."""
    assert _make().returncode == 0
    r = subprocess.run([os.path.join(HOST, "run_g2o_graph"), "600", "4", "100"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["nodes"] == 600
    assert res["chi2_after"] <= res["chi2_before"]
    assert res["chi2_after"] < 6 * 3000 * 3          # ~chi-square with 6E - 6N dof


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "_ref_test_g2o_graph")), reason="prebuilt reference driver not shipped")
def test_reference_driver_runs_unchanged(tmp_path):
    """The reference's own driver binary (built in the container, shipped like a prebuilt .so) on config 1."""
    env = dict(os.environ, FGO_SYNTH_POSES="1000", FGO_SYNTH_LOOKBACK="4", sr_start_frame="1", sr_end_frame="1001",
               gt_lookback_nodes="4", gt_optimize_step="250", gt_output_dir=str(tmp_path), sr_data_name="cfg1")
    r = subprocess.run([os.path.join(HOST, "_ref_test_g2o_graph")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stderr.splitlines() if "optimization error is" in l]
    assert len(lines) == 2
    before = float(lines[0].split()[-1]); after = float(lines[1].split()[-1])
    assert after < before
    assert os.path.exists(str(tmp_path / "cfg1_vo_after_trajectory_g2o.log"))



# ------------------------------------------------------------------------------------------------------------------
# GTSAM side: the reference's CGraphGT + imu_interface, compiled in place, driven by the reference's own drivers.

GT_SYMS = ["CGraphGT::CGraphGT()", "CGraphGT::firstNode(CCameraNode*, bool)", "CGraphGT::fakeOdoNode(CCameraNode*)",
           "CGraphGT::addNode(CCameraNode*)", "CGraphGT::optimizeGraph()", "CGraphGT::optimizeGraphBatch()", "CGraphGT::optimizeGraphIncremental()",
           "CGraphGT::addToGTSAM(MatchingResult&, bool)", "CGraphGT::addToGTSAM(gtsam::NavState&, int, bool)",
           "CGraphGT::addToGTSAM(CCameraNodeBA*, CCameraNodeBA*", "CGraphGT::bundleAdjust(MatchingResult*, CCameraNode*, CamModel*)",
           "CGraphGT::vroAdjust(MatchingResult*, CCameraNode*, CamModel*)", "CGraphGT::addPlaneFactor(CPlane*, int, int, double)",
           "CGraphGT::planeNodeAssociation(int, CPlaneNode*, double)", "CGraphGT::predictPlaneNode(",
           "CGraphGT::isSmallTrafo(MatchingResult&)", "CGraphGT::isLargeTrafo(MatchingResult&)", "CGraphGT::error()",
           "CGraphGT::camnodeSize()", "CGraphGT::writeG2O(", "CGraphGT::writeGTSAM(", "CGraphGT::writeTrajectory(", "CGraphGT::setWorld2Original(double)",
           "CGraphGT::setCamera2IMU(double)", "CGraphGT::setCamera2IMUTranslation(double, double, double)", "CGraphGT::initFromImu(double, double, double)",
           "CGraphGT::recordVROResult(MatchingResult&)", "CGraphGT::readVRORecord(", "CGraphGT::addNodeOffline(CCameraNode*, MatchingResult*, bool)",
           "CGraphGT::addEdgeOffline(MatchingResult*)", "CGraphGT::correctMatchingID(MatchingResult*)", "CGraphGT::trajectoryPLY(",
           "CImuBase::predictNext(int)", "CImuBase::predictNextFlag(double, gtsam::NavState&)", "CImuBase::findIndexAt(double)",
           "CImuBase::setStartPoint(double)", "CImuBase::getParam()", "CImuBase::predictBetween(int, int, gtsam::NavState&",
           "CImuVn100::readImuData(", "CImuVn100::getIMUParams()", "CImuMEMS::", "CGTParams::Instance()"]


def test_gtsam_side_library_builds_and_exports_surface():
    r = _make()
    assert r.returncode == 0, r.stderr[-2000:]
    if not os.path.exists(os.path.join(HOST, "libgtsam_graph.so")):
        pytest.skip("reference tree not present and no prebuilt wrapper library")
    out = subprocess.run(["nm", "-DC", os.path.join(HOST, "libgtsam_graph.so")], capture_output=True, text=True).stdout
    for sym in GT_SYMS:
        assert sym in out, "missing symbol " + sym


def test_no_reference_wrapper_source_in_repo():
    """the wrappers are compiled from the reference tree, not kept here in any form"""
    for f in ("g2o_graph.cpp", "g2o_graph.h", "gtsam_graph.cpp", "gtsam_graph.h", "imu_base.cpp", "imu_vn100.cpp", "g2o_parameter.cpp", "gt_parameter.cpp"):
        assert not os.path.exists(os.path.join(HOST, f)), f


def _so3_exp(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-8:
        return np.eye(3) + W + 0.5 * W @ W, np.eye(3) + 0.5 * W + W @ W / 6
    a, b, c = np.sin(th) / th, (1 - np.cos(th)) / th ** 2, (th - np.sin(th)) / th ** 3
    return np.eye(3) + a * W + b * W @ W, np.eye(3) + b * W + c * W @ W


def _rot_to_quat(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_quat()          # x y z w


def _quat_to_rot(q):
    from scipy.spatial.transform import Rotation
    return Rotation.from_quat(q).as_matrix()


def _make_logs(tmp_path, n_kf, vo_fail=()):
    assert _make("make_vio_logs").returncode == 0
    args = [os.path.join(HOST, "make_vio_logs"), str(tmp_path), str(n_kf), "3", "44"]
    if vo_fail:
        args.append(",".join(str(f) for f in vo_fail))
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]


def _driver_env(tmp_path, n_kf, name):
    return dict(os.environ, FGO_SYNTH_POSES=str(n_kf + 2), sr_start_frame="1", sr_end_frame="100000000", sr_data_name=name,
                imu_file=str(tmp_path / "imu.log"), imu_time_file=str(tmp_path / "img_time.log"),
                vro_results_file=str(tmp_path / "vro_results.log"), plane_aided="0", view_plane="0", chi2_for_vro="0",
                use_imu="1", gt_output_dir=str(tmp_path), trajectory_color="1")


def _rebuild_vio(tmp_path, n_kf, G, batch_at_end):
    """The graph the reference's offline VIO drivers build from the log files, rebuilt independently: numpy pose algebra +
    the C-ABI, ISAM2 step per record group exactly where the drivers call optimizeGraphIncremental
    (gtsam/test_vro_imu_graph.cpp:159-350).  Returns (error before the final step, error after, poses[n_kf, 7])."""
    def Rz(t): return np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1.0]])
    def Rx(t): return np.array([[1.0, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]])
    Ruc = Rz(np.pi / 2) @ Rx(np.pi / 2)                 # setCamera2IMU(0): RzRyRx(pi/2, 0, pi/2), zero translation
    Adj = np.zeros((6, 6)); Adj[:3, :3] = Ruc; Adj[3:, 3:] = Ruc
    XI, VI, BI = 0, 10 ** 6, 2 * 10 ** 6
    gr = G.Graph()
    gr.add_poses(np.array([[0, 0, 0, 0, 0, 0, 1.0]]), ids=[XI])
    w = np.zeros(21); w[[0, 6, 11, 15, 18, 20]] = 1e14                     # Diagonal::Sigmas(1e-7): gtsam_graph.cpp:338-341
    gr.add_prior(XI, np.array([0, 0, 0, 0, 0, 0, 1.0]), w)
    gr.add_vec3(VI, np.zeros(3)); gr.add_bias(BI, np.zeros(6))
    gr.add_prior_vec3(VI, np.zeros(3), 1e-3); gr.add_prior_bias(BI, np.zeros(6), 1e-3)
    imu = np.loadtxt(tmp_path / "imu.log").astype(np.float32).astype(np.float64)   # the reader parses floats (imu_vn100.cpp:86)
    imu_t = np.loadtxt(tmp_path / "imu.log")[:, 0]
    times = dict((int(a), b) for a, b in np.loadtxt(tmp_path / "img_time.log"))
    rec = np.loadtxt(tmp_path / "vro_results.log")
    start = int(np.argmin(np.abs(imu_t - times[1])))
    pim = G.Preintegrator()
    state = (np.array([0, 0, 0, 0, 0, 0, 1.0]), np.zeros(3), np.zeros(6))      # pose, velocity, bias the preintegration restarts from
    cur_frame, cur_node, k = 1, 0, 0
    node_of = {1: 0}
    while k < len(rec):
        row = rec[k]
        j, i = int(row[0]), int(row[1])
        def rel(row):
            Rr, V = _so3_exp(row[2:5])
            R = Ruc @ Rr @ Ruc.T
            t = Ruc @ (V @ row[5:8])
            Om = np.zeros((6, 6)); Om[np.triu_indices(6)] = row[8:29]; Om = Om + Om.T - np.diag(np.diag(Om))
            return R, t, (Adj @ Om @ Adj.T)[np.triu_indices(6)]
        if j > cur_frame:                                  # new frame: odometry edge + IMU factor + V / B values
            nid = len(node_of); node_of[j] = nid
            R, t, Om = rel(row)
            pi = gr.get_poses(ids=[XI + node_of[i]])[0]
            Ri = _quat_to_rot(pi[3:])
            pj = np.concatenate([Ri @ t + pi[:3], _rot_to_quat(Ri @ R)])
            gr.add_poses(np.array([pj]), ids=[XI + nid])
            gr.add_edges([XI + node_of[i]], [XI + nid], np.array([np.concatenate([t, _rot_to_quat(R)])]), np.array([Om]), tangent_order=G.FGO_TANGENT_GTSAM)
            for s in range(start + 40 * (j - 2), start + 40 * (j - 1)):
                pim.integrate(imu[s, 1:4], imu[s, 4:7], 0.005)
            xj, vj = pim.predict(state[0], state[1], state[2])
            gr.add_vec3(VI + nid, vj); gr.add_bias(BI + nid, np.zeros(6))     # addToGTSAM(NavState): V = prediction, B = *mp_prev_bias (never updated: zero)
            gr.add_imu([XI + nid - 1, VI + nid - 1, XI + nid, VI + nid, BI + nid - 1, BI + nid], pim.buf)
            cur_frame, cur_node = j, nid
            k += 1
        else:                                              # look-back records up to the next new frame
            while k < len(rec) and int(rec[k][0]) <= cur_frame:
                row = rec[k]
                R, t, Om = rel(row)
                gr.add_edges([XI + node_of[int(row[1])]], [XI + node_of[int(row[0])]], np.array([np.concatenate([t, _rot_to_quat(R)])]), np.array([Om]),
                             tangent_order=G.FGO_TANGENT_GTSAM)
                k += 1
        gr.isam2_update(0.1)                               # optimizeGraphIncremental after every loop iteration
        vals = gr.get_poses(ids=[XI + cur_node, VI + cur_node, BI + cur_node])
        state = (vals[0], vals[1][:3].copy(), vals[2][:6].copy())
        pim.reset(state[2])
    e0 = gr.error()
    e1 = e0
    if batch_at_end:
        gr.optimize_gtsam(100)
        e1 = gr.error()
    return e0, e1, gr.get_poses(ids=[XI + q for q in range(n_kf)])


def _errors(stderr):
    lines = [l for l in stderr.splitlines() if "optimization error is" in l]
    return [float(l.split()[-1]) for l in lines]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "_ref_test_vro_imu_graph")), reason="prebuilt reference driver not shipped")
def test_reference_vio_driver_runs_unchanged(tmp_path):
    """gtsam/test_vro_imu_graph.cpp, compiled in place, on synthetic VRO + VN100 logs (BASELINE config 4's harness at a
    size the per-record ISAM2 flow finishes quickly): ISAM2 (fgo_isam2_update) after every record, error() and the
    trajectory writers at the end.  Cross-check: the same graph rebuilt by an independent numpy implementation through
    the C-ABI with the ISAM2 steps at the same places -- error to 1e-6 relative, poses to 1e-6."""
    import graph_slam_amd as G
    n_kf = 80
    _make_logs(tmp_path, n_kf)
    r = subprocess.run([os.path.join(HOST, "_ref_test_vro_imu_graph")], capture_output=True, text=True, timeout=900, env=_driver_env(tmp_path, n_kf, "vio"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    errs = _errors(r.stderr)
    assert len(errs) == 1, r.stderr[-1500:]
    traj = np.loadtxt(tmp_path / "vio_vio_trajectory.log")
    truth = np.loadtxt(tmp_path / "truth.log")
    assert traj.shape == (n_kf, 9)
    assert (traj[:, 8] == np.arange(1, n_kf + 1)).all()                      # seq ids = frame ids
    assert np.abs(traj[:, 1:4] - truth[:, 1:4]).max() < 0.1
    assert (tmp_path / "vio_vio.ply").read_text().startswith("ply")
    e0, _, poses = _rebuild_vio(tmp_path, n_kf, G, batch_at_end=False)
    print("reference VIO driver: error %.9e, independent rebuild %.9e" % (errs[0], e0))
    assert abs(e0 - errs[0]) <= 1e-6 * max(errs[0], 1e-9), (e0, errs[0])
    assert np.abs(traj[:, 1:4] - poses[:, :3]).max() < 1e-6                  # mp_w2o is the identity in this driver


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "_ref_test_ba_imu_graph")), reason="prebuilt reference driver not shipped")
def test_reference_ba_imu_driver_runs_unchanged(tmp_path):
    """gtsam/test_ba_imu_graph.cpp (named by BASELINE.json's north star), compiled in place: same replay with use_imu,
    ISAM2 per record, then error(), optimizeGraphBatch() (fgo_optimize_gtsam) and error() again (:448-453)."""
    import graph_slam_amd as G
    n_kf = 60
    _make_logs(tmp_path, n_kf)
    r = subprocess.run([os.path.join(HOST, "_ref_test_ba_imu_graph")], capture_output=True, text=True, timeout=900, env=_driver_env(tmp_path, n_kf, "ba"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    errs = _errors(r.stderr)
    assert len(errs) == 2, r.stderr[-1500:]
    assert errs[1] <= errs[0] * (1 + 1e-9)
    e0, e1, poses = _rebuild_vio(tmp_path, n_kf, G, batch_at_end=True)
    print("reference BA+IMU driver: error %.9e -> %.9e, independent rebuild %.9e -> %.9e" % (errs[0], errs[1], e0, e1))
    assert abs(e0 - errs[0]) <= 1e-6 * max(errs[0], 1e-9), (e0, errs[0])
    assert abs(e1 - errs[1]) <= 1e-5 * max(errs[1], 1e-9), (e1, errs[1])
    logs = [f for f in os.listdir(tmp_path) if f.endswith("trajectory.log")]
    assert logs, os.listdir(tmp_path)


def _read_graph_dump(path):
    """FGO_GRAPH_DUMP (host/gtsam_bridge.cpp): values and factor descriptors of the graph an error() call was evaluated on"""
    vals, facs, err = {}, [], None
    for line in open(path):
        w = line.split()
        if w[0] == "E":
            err = float(w[1])
        elif w[0] == "V":
            vals[w[1] + w[2]] = (int(w[3]), np.array([float(x) for x in w[4:11]]))
        elif w[0] == "F":
            nk = int(w[2])
            facs.append((int(w[1]), w[3:3 + nk], np.array([float(x) for x in w[3 + nk:]])))
    return err, vals, facs


def _independent_error(vals, facs, calib=None, bps=None):
    """0.5 * sum of squared whitened residuals of a dumped graph, factor by factor with the oracle's factor functions
    (kinds: host/shim/gtsam_lite.h FactorDesc::Kind); reprojection factors (numpy: Cal3DS2 `calib`, body_P_sensor `bps`)"""
    from graph_slam_amd import scenarios as S
    import graph_slam_amd as G
    from tests import orc_binding as orc
    def ut(u, n):
        W = np.zeros((n, n)); W[np.triu_indices(n)] = u; return W + W.T - np.diag(np.diag(W))
    tot, count = 0.0, {}
    pim = orc.Preint(np.zeros(6), np.zeros((0, 3)), np.zeros((0, 3)), 0.005)
    for kind, keys, pl in facs:
        x = [vals[k][1] for k in keys]
        if kind == 0:                                  # PRIOR_POSE: t q info21
            e = orc.prior(x[0], pl[:7], jac=False); tot += 0.5 * e @ ut(pl[7:28], 6) @ e
        elif kind in (1, 2, 3):                        # PRIOR_VEC3 / BIAS / POINT: v6 sigma
            dim = 6 if kind == 2 else 3
            tot += 0.5 * np.sum((x[0][:dim] - pl[:dim]) ** 2) / pl[6] ** 2
        elif kind == 4:                                # BETWEEN
            e = orc.between(x[0], x[1], pl[:7], jac=False); tot += 0.5 * e @ ut(pl[7:28], 6) @ e
        elif kind == 5:                                # IMU: gravity(3) payload(287)
            pim.buf[:] = pl[3:3 + 287]
            r = pim.factor(x[0], x[1][:3], x[2], x[3][:3], x[4][:6], x[5][:6], jac=False, g=pl[:3])
            tot += 0.5 * r @ G.preint_information(pl[3:3 + 287]) @ r
        elif kind == 6:                                # PLANE: z(4) cov_ut6
            r = orc.plane_factor(x[0], x[1][:4], orc.plane(*pl[:4]), jac=False)
            tot += 0.5 * r @ np.linalg.inv(ut(pl[4:10], 3)) @ r
        elif kind == 7:                                # REPROJ: u v sigma; keys pose, point
            X, pw = x[0], x[1][:3]
            cq = S._quat_mul(X[None, 3:], np.asarray(bps)[None, 3:])[0]
            ct = X[:3] + S._quat_rot(X[None, 3:], np.asarray(bps)[None, :3])[0]
            pk = S._quat_rot(cq[None] * np.array([-1, -1, -1, 1.0]), (pw - ct)[None])
            r = S._project(pk, np.asarray(calib))[0] - pl[:2] if pk[0, 2] > 0 else np.full(2, 2 * calib[0])
            tot += 0.5 * float(r @ r) / pl[2] ** 2
        else:
            raise AssertionError("unexpected factor kind %d" % kind)
        count[kind] = count.get(kind, 0) + 1
    return tot, count


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "_ref_test_vro_imu_graph")), reason="prebuilt reference driver not shipped")
def test_reference_vio_driver_plane_aided(tmp_path):
    """BASELINE config 4's third leg through the reference's OWN code (VERDICT r2 missing #2): plane_aided = 1.  VRO 'fails' on
    a few frames (void records), so gtsam/test_vro_imu_graph.cpp:202-314 runs its plane branch: CPlaneNode::extractPlanes (the
    stand-in segments the synthetic room's range image), CGraphGT::planeNodeAssociation (:1346-1503), predictPlaneNode
    (:877-1100: projection of the previous planes' pixels, region growing on the range image, plane refit), addPlaneFactor
    (:1118-1298: covariance through Unit3 bases / OrientedPlane3::transform Jacobians, Gaussian::Covariance) -- all compiled
    in place, unmodified.  Checks: OrientedPlane3Factors reach libfgo; the error the driver prints equals an INDEPENDENT
    evaluation of every factor of the graph it built (oracle factor functions on the dumped descriptors) to 1e-9; the plane
    landmarks are the room's walls; the trajectory stays on the truth."""
    n_kf, fail = 60, (20, 21, 35, 36, 50)
    _make_logs(tmp_path, n_kf, vo_fail=fail)
    env = _driver_env(tmp_path, n_kf, "pvio")
    env.update(plane_aided="1", FGO_SYNTH_TRUTH=str(tmp_path / "truth.log"), FGO_GRAPH_DUMP=str(tmp_path / "graph_dump.txt"))
    r = subprocess.run([os.path.join(HOST, "_ref_test_vro_imu_graph")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    errs = _errors(r.stderr)
    assert len(errs) == 1, r.stderr[-1500:]
    err_dump, vals, facs = _read_graph_dump(tmp_path / "graph_dump.txt")
    assert abs(err_dump - errs[0]) <= 1e-6 * max(errs[0], 1e-9)              # (the log line prints 6 decimals)
    ind, count = _independent_error(vals, facs)
    print("plane-aided VIO driver: error %.9e, independent evaluation %.9e, factors by kind %s" % (err_dump, ind, count))
    if os.environ.get("FGO_TEST_VERBOSE"):
        print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "lane" in l or "landmark" in l)[-6000:])
    # plane factors: the first node's wall (firstPlaneNode), one per VO failure from the association of the previous node's
    # planes (new landmarks: potentialPlaneNodes only looks a few nodes back), and the propagated ones that pass the
    # reference's 70 % overlap test (predictPlaneNode)
    assert count.get(6, 0) >= len(fail) + 1, count
    assert "detect a plane by plane propagation" in r.stderr + r.stdout
    assert count.get(5, 0) == n_kf - 1 and count.get(4, 0) > n_kf
    assert abs(ind - err_dump) <= 1e-9 * max(err_dump, 1e-9), (ind, err_dump)
    # the plane landmarks are walls of the room: axis-aligned unit normals, distance = the wall's offset
    truth = np.loadtxt(tmp_path / "truth.log")
    lo, hi = truth[:, 1:4].min(0) - 2.0, truth[:, 1:4].max(0) + 2.0
    planes = [v[1][:4] for k, v in vals.items() if k.startswith("l")]
    assert len(planes) >= 2
    for pl in planes:
        a = int(np.argmax(np.abs(pl[:3])))
        assert abs(abs(pl[a]) - 1) < 2e-3, pl
        wall = -pl[3] / pl[a]                                                # n . p + d = 0  ->  p_a = -d / n_a
        assert min(abs(wall - lo[a]), abs(wall - hi[a])) < 0.03, (pl, lo, hi)
    traj = np.loadtxt([f for f in (tmp_path / f for f in os.listdir(tmp_path)) if str(f).endswith("trajectory.log")][0])
    assert traj.shape[0] == n_kf and np.abs(traj[:, 1:4] - truth[:, 1:4]).max() < 0.1


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "run_bundle_adjust")), reason="prebuilt harness not shipped")
def test_reference_bundle_adjust_two_view():
    """CGraphGT::bundleAdjust (gtsam/gtsam_graph.cpp:500-610, the reference's own code compiled in place): two-view BA with
    Cal3DS2 reprojection factors + point priors through LevenbergMarquardtOptimizer (fgo_optimize_gtsam), then
    Marginals::marginalCovariance of the second camera (fgo_marginal_cov) inverted into the edge's information matrix.
    The recovered relative pose must match the pose the synthetic features were generated with (noise: 0.3 px, 5-10 mm)."""
    r = subprocess.run([os.path.join(HOST, "run_bundle_adjust"), "80", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["max_abs_dt"] < 0.01 and d["max_abs_dR"] < 0.005, d
    assert min(d["info_diag"]) > 0 and d["info_sym_err"] < 1e-6 * max(d["info_diag"]), d


@pytest.mark.gpu
def test_g2o_file_load_optimise_save_roundtrip(tmp_path):
    """.g2o text in (SparseOptimizer::load, incl. FIX), the reference's schedule on the GPU, .g2o text out; chi2 before /
    after equal to the same graph fed through the C-ABI, and the saved file loads back to the optimised chi2"""
    import graph_slam_amd as G
    assert _make("g2o_file_tool").returncode == 0
    n = 400
    g = G.synth_manhattan3d(n, 4, 2, seed=9)
    path = tmp_path / "in.g2o"
    with open(path, "w") as fh:
        for k in range(n):
            fh.write("VERTEX_SE3:QUAT %d %s\n" % (k, " ".join("%.17g" % v for v in g["poses"][k])))
        fh.write("FIX 0\n")
        fh.write("# a comment line and an unknown tag are skipped\nPARAMS_SE3OFFSET 0 0 0 0 0 0 0 1\n")
        for e in range(len(g["ei"])):
            fh.write("EDGE_SE3:QUAT %d %d %s %s\n" % (g["ei"][e], g["ej"][e], " ".join("%.17g" % v for v in g["meas"][e]),
                                                    " ".join("%.17g" % v for v in g["info"][e])))
    out = tmp_path / "out.g2o"
    r = subprocess.run([os.path.join(HOST, "g2o_file_tool"), str(path), "6", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr[-1500:])
    w = r.stdout.split()
    assert int(w[1]) == n and int(w[3]) == len(g["ei"]) and int(w[5]) == 6
    c0, c1 = float(w[8]), float(w[10])
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    gr = G.Graph(); gr.add_poses(g["poses"], fixed); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    assert abs(gr.chi2() - c0) <= 1e-12 * c0
    for _ in range(3):
        gr.optimize(2)
    assert abs(gr.chi2() - c1) <= 1e-9 * c1
    r2 = subprocess.run([os.path.join(HOST, "g2o_file_tool"), str(out), "0"], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0 and abs(float(r2.stdout.split()[8]) - c1) <= 1e-9 * c1
    assert "FIX 0" in out.read_text()


@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "run_imu_mems")), reason="prebuilt harness not shipped")
def test_reference_mems_imu_interface(tmp_path):
    """VERDICT r2 missing #7: the reference's CImuMEMS (gtsam/imu_MEMS.cpp, compiled in place) -- integer log `id1 gx gy gz ax ay az id2`,
    counts -> rad/s (Gi2V, through a float) and m/s^2 (Ai2V), synchronisation point where id1 restarts at 1, prior bias = mean of the
    stationary samples before it (gravity 9.81 removed from az), dt = 10 ms (a float), MakeSharedD(9.81), its own noise parameters --
    against a numpy restatement of the parsing and the ORACLE's preintegration / prediction with those parameters.  Host-only."""
    from tests import orc_binding as orc
    rng = np.random.default_rng(0)
    n0, n1 = 120, 300
    rows = []
    for k in range(n0):
        g = rng.integers(-3, 4, 3); a = [rng.integers(-4, 5), rng.integers(-4, 5), -397 + rng.integers(-3, 4)]
        rows.append((k + 5, g[0], g[1], g[2], a[0], a[1], a[2], 0))
    for k in range(n1):
        t = k * 0.01
        g = np.round([120 * np.sin(1.3 * t), -80 * np.cos(0.7 * t), 60 * np.sin(2.1 * t + 1)]).astype(int)
        a = np.round([40 * np.sin(0.9 * t), 30 * np.cos(1.7 * t), -397 + 20 * np.sin(1.1 * t)]).astype(int)
        rows.append((k + 1, g[0], g[1], g[2], a[0], a[1], a[2], k // 3 + 1))
    log = tmp_path / "mems.log"
    log.write_text("\n".join(" ".join(str(int(v)) for v in r) for r in rows))          # no trailing newline: the reader loops on eof()
    r = subprocess.run([os.path.join(HOST, "run_imu_mems"), str(log), "200", "200", "260"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    raw = np.array(rows, dtype=np.int64)
    gyro = (np.float32(raw[:, 1:4] * 80.0).astype(np.float64) / 1092.0) * np.pi / 180.0        # Gi2V: the product goes through a float
    acc = raw[:, 4:7] * 0.002522 * 9.81                                                     # Ai2V
    assert d["n"] == n0 + n1 and d["syn_start"] == n0
    np.testing.assert_allclose(d["first"], np.concatenate([gyro[0], acc[0]]), rtol=1e-15)
    bias = np.concatenate([acc[:n0].mean(0) + [0, 0, 9.81], gyro[:n0].mean(0)])             # ConstantBias(acc, gyro)
    np.testing.assert_allclose(d["bias"], bias, rtol=1e-12, atol=1e-15)
    dt = float(np.float32(0.01))
    assert d["dt"] == dt
    gs, as_ = (np.pi / 180 * 3.6) / 60, 0.1 / 60                                            # imu_MEMS.cpp:19-20
    var = [as_ ** 2, gs ** 2, 1e-4, 1e-8, 1e-8, 1e-5]
    pim = orc.Preint(bias, acc[n0:n0 + 200], gyro[n0:n0 + 200], dt, variances=var)
    pay = np.array(d["payload"])
    np.testing.assert_allclose(pay[:62], pim.buf[:62], rtol=0, atol=1e-12)                  # dt, deltas, bias Jacobians, bias
    np.testing.assert_allclose(pay[62:], pim.buf[62:], rtol=1e-10, atol=1e-10 * np.abs(pim.buf[62:]).max())   # preintMeasCov with the MEMS noise model
    g981 = np.array([0, 0, 9.81])
    xj, vj = pim.predict(np.array([0, 0, 0, 0, 0, 0, 1.0]), np.zeros(3), bias, g=g981)
    np.testing.assert_allclose(d["predict_next"][:7], xj, atol=1e-11); np.testing.assert_allclose(d["predict_next"][7:], vj, atol=1e-11)
    pim2 = orc.Preint(bias, acc[n0 + 200:n0 + 260], gyro[n0 + 200:n0 + 260], dt, variances=var)
    xk, vk = pim2.predict(xj, vj, bias, g=g981)
    np.testing.assert_allclose(d["predict_between"][:7], xk, atol=1e-10); np.testing.assert_allclose(d["predict_between"][7:], vk, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "run_ba_multiframe")), reason="prebuilt harness not shipped")
def test_reference_multiframe_ba_builder(tmp_path):
    """VERDICT r2 missing #3: the reference's OWN multi-frame bundle-adjustment builder, CGraphGT::addToGTSAM(CCameraNodeBA*,
    CCameraNodeBA*, matches, CamModel*) (gtsam/gtsam_graph.cpp:370-448, compiled in place; its call sites in the reference's
    drivers are commented out, host/examples/run_ba_multiframe.cpp issues them in the same pattern) on 60 synthetic keyframes:
    landmark ids carried from keyframe to keyframe, PriorFactor<Point3> + GenericProjectionFactor<Pose3, Point3, Cal3DS2> with
    body_P_sensor = camera-to-IMU, VO BetweenFactors from addNodeOffline, LevenbergMarquardtOptimizer -> libfgo, which eliminates
    the (> 1000) landmarks first (kernels_ba.hip).  Checks: the error before and after the optimisation equals an INDEPENDENT
    evaluation of every factor of the dumped graph (oracle factor functions + a numpy reprojection) to 1e-9; the optimised error
    is at the noise floor; the keyframe poses land on the truth."""
    pre = str(tmp_path / "dump")
    env = dict(os.environ, FGO_GRAPH_DUMP_PREFIX=pre, LD_LIBRARY_PATH=HOST + ":" + os.path.dirname(HOST) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([os.path.join(HOST, "run_ba_multiframe"), "60", "6000", "3", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["landmarks"] >= 1000 and d["keyframes"] == 60
    for tag, key in (("0", "error0"), ("1", "error1")):
        err, vals, facs = _read_graph_dump(pre + tag + ".txt")
        ind, count = _independent_error(vals, facs, calib=d["calib"], bps=d["body_P_sensor"])
        assert abs(err - d[key]) <= 1e-12 * d[key]
        assert abs(ind - err) <= 1e-9 * err, (tag, ind, err)
        assert count[3] == d["landmarks"] and count[4] == 59 and count[7] >= 2 * d["landmarks"]       # point priors, VO factors, projections
    n_obs = count[7]
    print("multi-frame BA through the reference's builder: %d landmarks, %d projection factors, error %.4e -> %.4e" % (d["landmarks"], n_obs, d["error0"], d["error1"]))
    assert d["error1"] < 0.2 * d["error0"]
    assert d["error1"] < 1.5 * n_obs                            # 2 residuals per observation at sigma 1 px with 0.3 px noise + priors: well below n_obs
    assert d["max_abs_dt"] < 0.06 and d["max_abs_dR"] < 0.03          # (3 m of path; landmark priors at the noisy first sightings pull at the per cent level)
