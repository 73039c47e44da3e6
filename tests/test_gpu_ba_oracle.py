"""GPU parity of the bundle-adjustment kernels that run at BASELINE.json config 3 -- landmarks eliminated first
(csrc/kernels_ba.hip: k_ba_linearize, k_ba_cameras, k_ba_points, k_ba_schur, k_ba_back; on at >= 1000 eligible landmarks) --
DIRECTLY against the oracle (until now they were only compared with the generic GPU path).
Reference: CGraphGT::addToGTSAM(CCameraNodeBA*, ...) gtsam/gtsam_graph.cpp:370-448 (Cal3DS2, PriorFactor<Point3> sigma
0.014, GenericProjectionFactor sigma 1 px, body_P_sensor) solved by LevenbergMarquardtOptimizer (:1784-1788).

  * the LM run of the oracle (GTSAM semantics, landmarks as ordinary columns of its sparse Cholesky) on a 300-keyframe /
    8 000-landmark graph: iterations, trials, lambda trajectory 1e-12, error trajectory 1e-8, estimate 1e-7;
  * the reduced camera system S = H_cc - W (H_pp + lambda I)^-1 W^T, g = b_c - W (H_pp + lambda I)^-1 b_p the device
    forms, against the Schur complement of the ORACLE's dense H / b (numpy), 1e-10 of the largest entry;
  * the same in a mixed VIO + BA graph (IMU factors, planes, 1 300 landmarks);
  * fgo_isam2_update on a context whose structure has the landmarks eliminated; fgo_linearize's n_free.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
import graph_slam_amd.scenarios as S
from tests import orc_binding as orc
from tests.util import info_ut, vio_graph, mixed_oracle


def ba_oracle(p, pose_prior_sigma=1e-3, odometry_sigma=(0.002, 0.005)):
    """the graph scenarios.ba_graph assembles through the C-ABI, handed to the oracle (same factor order)"""
    n_kf, n_pts = len(p["poses0"]), len(p["points0"])
    N = n_kf + n_pts
    values = np.zeros((N, 7)); values[:n_kf] = p["poses0"]; values[n_kf:, :3] = p["points0"]
    vkind = np.zeros(N, np.int32); vkind[n_kf:] = orc.VK_POINT
    no = len(p["obs_uv"])
    meas = np.zeros((no, 7)); meas[:, :2] = p["obs_uv"]
    info = np.zeros((no, 21)); info[:, 0] = 1.0 / p["pixel_sigma"] ** 2
    a, b = p["poses"][:-1], p["poses"][1:]
    qa_c = a[:, 3:] * np.array([-1, -1, -1, 1.0])
    zt = S._quat_rot(qa_c, b[:, :3] - a[:, :3]); zq = S._quat_mul(qa_c, b[:, 3:])
    wi = np.zeros(21); wi[[0, 6, 11]] = 1.0 / odometry_sigma[0] ** 2; wi[[15, 18, 20]] = 1.0 / odometry_sigma[1] ** 2
    ei = np.concatenate([p["obs_kf"], np.arange(n_kf - 1)]).astype(np.int32)
    ej = np.concatenate([n_kf + p["obs_pt"], np.arange(1, n_kf)]).astype(np.int32)
    kind = np.concatenate([np.full(no, orc.FK_REPROJ), np.full(n_kf - 1, orc.FK_BETWEEN)]).astype(np.int32)
    po = orc.Problem(values, np.zeros(N, np.uint8), ei, ej, np.concatenate([meas, np.concatenate([zt, zq], 1)]),
                     np.concatenate([info, np.tile(wi, (n_kf - 1, 1))]))
    po.set_kinds(vkind, kind)
    po.set_calibration(np.array(p["calib"], np.float64), p["bps"])
    w = np.zeros(21); w[[0, 6, 11, 15, 18, 20]] = 1.0 / pose_prior_sigma ** 2
    pw6 = np.zeros((6, 6)); pw6[:3, :3] = np.eye(3) / p["point_sigma"] ** 2
    ids = np.concatenate([[0], np.arange(n_kf, N)]).astype(np.int32)
    mean = np.zeros((1 + n_pts, 7)); mean[0] = p["poses"][0]; mean[1:, :3] = p["points0"]
    infos = np.tile(info_ut(pw6), (1 + n_pts, 1)); infos[0] = w
    po.add_priors(ids, mean, infos)
    return po


def _check_lm(gr, po, n_cam, iters=20):
    e0g, e0o = gr.error(), po.error_gtsam()
    assert abs(e0g - e0o) <= 1e-11 * e0o
    rg, sg = gr.optimize_gtsam(iters)
    ro, so = po.optimize_gtsam(iters)
    assert rg == ro and sg.iterations == so.iterations and sg.trials == so.trials, (rg, ro, sg.trials, so.trials)
    tg, to = gr.trace(), po.trace()
    np.testing.assert_allclose(tg[1], to[1], rtol=1e-12)                       # lambda trajectory
    np.testing.assert_allclose(tg[0], to[0], rtol=1e-8)                        # error trajectory
    assert abs(gr.error() - po.error_gtsam()) <= 1e-8 * po.error_gtsam()
    V, Vo = gr.get_poses(), po.get_poses()
    sgn = np.sign(np.sum(V[:n_cam, 3:] * Vo[:n_cam, 3:], axis=1))[:, None]
    assert np.abs(V[:n_cam, :3] - Vo[:n_cam, :3]).max() < 1e-7
    assert np.abs(V[:n_cam, 3:] * sgn - Vo[:n_cam, 3:]).max() < 1e-7
    return sg, V, Vo


@pytest.mark.parametrize("start_noise", [0.0, 0.04])
def test_landmark_elimination_lm_vs_oracle(start_noise):
    """300 keyframes / 8 000 landmarks / ~77 000 observations through the kernels config 3 runs (the structure has the
    landmarks eliminated: fgo_debug_read_reduced answers) against the oracle's LM on the same graph"""
    p = S.ba_problem(300, 8000)
    if start_noise > 0:                                                        # a rougher start: more iterations, rejected trials
        rng = np.random.default_rng(5)
        p["poses0"] = p["poses0"].copy(); p["poses0"][1:, :3] += rng.normal(size=(299, 3)) * start_noise
        p["points0"] = p["points0"] + rng.normal(size=p["points0"].shape) * start_noise
    gr, po = S.ba_graph(p), ba_oracle(p)
    n = G.C.c_int64()
    gr._chk(G.lib.fgo_debug_read_reduced(gr._h, 0.0, None, None, G.C.byref(n)))   # FGO_ESTATE unless the landmarks are eliminated
    assert n.value == 300 + int((np.bincount(p["obs_pt"], minlength=8000) == 0).sum())
    sg, V, Vo = _check_lm(gr, po, 300)
    assert sg.n_free == 300 + 8000
    assert np.abs(V[300:, :3] - Vo[300:, :3]).max() < 1e-7                     # the eliminated landmarks (k_ba_back)
    assert gr.error() < 0.9 * 0.5 * sg.chi2_initial


def _rows(base, n_pts, seen):
    """scalar rows of the oracle's dense system (6 per variable, ids ascending): `keep` = every variable that stays a column
    (the non-landmark variables and landmarks nobody observes), `lm3` = the 3 real rows of every eliminated landmark"""
    elim = np.nonzero(seen > 0)[0]
    stay = np.concatenate([np.arange(base), base + np.nonzero(seen == 0)[0]])
    keep = (6 * stay[:, None] + np.arange(6)[None, :]).ravel()
    lm3 = (6 * (base + elim)[:, None] + np.arange(3)[None, :]).ravel()
    return keep, lm3


def _schur_of_oracle(Ho, bo, cam, lm3, lam):
    """dense Schur complement of the oracle's system onto the rows `cam`, eliminating the scalar rows `lm3` (damped)"""
    Hcc, Hcp, Hpp = Ho[np.ix_(cam, cam)], Ho[np.ix_(cam, lm3)], Ho[np.ix_(lm3, lm3)] + lam * np.eye(len(lm3))
    X = np.linalg.solve(Hpp, np.concatenate([Hcp.T, bo[lm3, None]], 1))
    return Hcc - Hcp @ X[:, :-1], bo[cam] - Hcp @ X[:, -1]


@pytest.mark.parametrize("lam", [0.0, 1e-5, 3.0])
def test_reduced_camera_system_vs_oracle_schur_complement(lam):
    """fgo_linearize-level check of what k_ba_linearize / k_ba_cameras / k_ba_points / k_ba_schur produce: the reduced
    system the block Cholesky factors, against numpy's Schur complement of the oracle's dense H, b"""
    n_kf, n_pts = 40, 1100
    p = S.ba_problem(n_kf, n_pts)
    gr, po = S.ba_graph(p), ba_oracle(p)
    Sg, gg = gr.read_reduced(lam)
    Ho, bo = po.dense_system()
    assert Ho.shape[0] == 6 * (n_kf + n_pts)
    cam, lm3 = _rows(n_kf, n_pts, np.bincount(p["obs_pt"], minlength=n_pts))
    assert Sg.shape[0] == len(cam)
    So, go = _schur_of_oracle(Ho, bo, cam, lm3, lam)
    # pose 0 carries the 1e6 prior only (sigma 1e-3), nothing like the 1e14 of the VIO graphs: one scale for all entries
    np.testing.assert_allclose(Sg, So, rtol=0, atol=1e-10 * np.abs(So).max())
    np.testing.assert_allclose(gg, go, rtol=0, atol=1e-10 * np.abs(go).max())
    assert abs(gr.chi2() - po.chi2()) <= 1e-11 * po.chi2()


def test_reduced_system_with_wide_covisibility():
    """every landmark seen by up to 100 keyframes: the first cameras have more than 80 row blocks in the reduced system, which the
    per-camera kernel (k_ba_schur_cam: 80 accumulator slots) leaves to the block-by-block kernel -- both kernels fill ONE reduced
    system; against the oracle's Schur complement as above"""
    n_kf, n_pts = 100, 1200
    p = S.ba_problem(n_kf, n_pts, obs_per_pt=100)
    seen = np.bincount(p["obs_pt"], minlength=n_pts)
    cov = np.zeros((n_kf, n_kf), bool)
    for j in range(n_pts):
        k = p["obs_kf"][p["obs_pt"] == j]
        cov[np.ix_(k, k)] = True
    rows_of_first = int(np.triu(cov)[0].sum())
    assert rows_of_first > 80 and int(np.triu(cov)[-1].sum()) <= 80, rows_of_first     # both kernels take part
    gr, po = S.ba_graph(p), ba_oracle(p)
    Sg, gg = gr.read_reduced(1e-4)
    Ho, bo = po.dense_system()
    cam, lm3 = _rows(n_kf, n_pts, seen)
    assert Sg.shape[0] == len(cam)
    So, go = _schur_of_oracle(Ho, bo, cam, lm3, 1e-4)
    np.testing.assert_allclose(Sg, So, rtol=0, atol=1e-10 * np.abs(So).max())
    np.testing.assert_allclose(gg, go, rtol=0, atol=1e-10 * np.abs(go).max())


def _vio_ba(seed=3, n_kf=40, n_pts=1300):
    """tests/util.vio_graph (poses, velocities, biases, IMU + between + plane factors, priors) plus Point3 landmarks seen
    through the SR4000 model by up to 6 neighbouring keyframes each; every 40th landmark by one keyframe only"""
    rng = np.random.default_rng(seed)
    g = vio_graph(rng, n_kf=n_kf, with_planes=True)
    g["imu_info"] = np.array([G.preint_information(q.buf) for q in g["imu_pre"]])     # same information on both sides
    K, npl = g["n_kf"], g["n_planes"]
    X = g["truth_X"]
    bps = np.array([0.0, 0.0, 0.0, 0.5, 0.5, 0.5, 0.5])
    cam_q = S._quat_mul(X[:, 3:], np.broadcast_to(bps[3:], (K, 4)))
    cam_t = X[:, :3]
    centre = rng.integers(0, K, n_pts)
    pc = np.stack([rng.uniform(-0.3, 0.3, n_pts), rng.uniform(-0.25, 0.25, n_pts), np.ones(n_pts)], 1) * rng.uniform(2.0, 5.0, (n_pts, 1))
    pw = cam_t[centre] + S._quat_rot(cam_q[centre], pc)
    base = 3 * K + npl
    ei, ej, meas = [], [], []
    for j in range(n_pts):
        span = 1 if j % 40 == 0 else 6
        for k in range(max(0, centre[j] - span // 2), min(K, centre[j] - span // 2 + span)):
            pk = S._quat_rot(cam_q[k][None] * np.array([-1, -1, -1, 1.0]), (pw[j] - cam_t[k])[None])
            if pk[0, 2] > 0.5 and abs(pk[0, 0] / pk[0, 2]) < 0.45 and abs(pk[0, 1] / pk[0, 2]) < 0.45:
                m = np.zeros(7); m[:2] = S._project(pk, S.SR4000)[0] + rng.normal(size=2)
                ei.append(k); ej.append(base + j); meas.append(m)
    no = len(ei)
    vals = np.zeros((n_pts, 7)); vals[:, :3] = pw + rng.normal(size=pw.shape) * 0.014
    g["values"] = np.concatenate([g["values"], vals]); g["vkind"] = np.concatenate([g["vkind"], np.full(n_pts, orc.VK_POINT, np.int32)])
    w = np.zeros((no, 21)); w[:, 0] = 1.0
    g["ei"] = np.concatenate([g["ei"], np.array(ei, np.int32)]); g["ej"] = np.concatenate([g["ej"], np.array(ej, np.int32)])
    g["kind"] = np.concatenate([g["kind"], np.full(no, orc.FK_REPROJ, np.int32)])
    g["meas"] = np.concatenate([g["meas"], np.array(meas)]); g["info"] = np.concatenate([g["info"], w])
    pw6 = np.zeros((6, 6)); pw6[:3, :3] = np.eye(3) / 0.014 ** 2
    g["prior_ids"] = np.concatenate([g["prior_ids"], np.arange(base, base + n_pts, dtype=np.int32)])
    g["prior_mean"] = np.concatenate([g["prior_mean"], vals]); g["prior_info"] = np.concatenate([g["prior_info"], np.tile(info_ut(pw6), (n_pts, 1))])
    g["bps"] = bps; g["n_points"] = n_pts; g["point_base"] = base
    return g


def _vio_ba_gpu(g):
    from tests.test_gpu_imu import vio_gpu
    K, npl, n_pts, base = g["n_kf"], g["n_planes"], g["n_points"], g["point_base"]
    sub = dict(g)
    keep = g["kind"] != orc.FK_REPROJ
    for key in ("ei", "ej", "kind", "meas", "info"):
        sub[key] = g[key][keep]
    gr = vio_gpu(sub)                                                          # poses, V, B, planes, between / plane / IMU factors, the 3 priors
    pid = np.arange(base, base + n_pts, dtype=np.int64)
    pts = np.ascontiguousarray(g["values"][base:, :3])
    gr._chk(G.lib.fgo_add_points3(gr._h, n_pts, S._i64p(pid), S._dp(pts), 0.014))
    gr.set_calibration(g["calib"], g["bps"])
    r = ~keep
    okf = np.ascontiguousarray(g["ei"][r], np.int64); opt = np.ascontiguousarray(g["ej"][r], np.int64)
    uv = np.ascontiguousarray(g["meas"][r, :2])
    gr._chk(G.lib.fgo_add_reprojs(gr._h, len(uv), S._i64p(okf), S._i64p(opt), S._dp(uv), 1.0))
    return gr


def test_mixed_vio_ba_graph_vs_oracle():
    """IMU + between + plane factors + 1 300 landmarks (eliminated) in ONE graph: start error, the reduced system of all
    non-landmark variables (poses, velocities, biases, planes) against the oracle's Schur complement, then the LM run"""
    g = _vio_ba()
    K, npl, n_pts, base = g["n_kf"], g["n_planes"], g["n_points"], g["point_base"]
    gr, po = _vio_ba_gpu(g), mixed_oracle(g)
    Sg, gg = gr.read_reduced(1e-5)
    Ho, bo = po.dense_system()
    seen = np.bincount(g["ej"][g["kind"] == orc.FK_REPROJ] - base, minlength=n_pts)
    assert (seen > 0).sum() >= 1000
    cam, lm3 = _rows(base, n_pts, seen)
    assert Sg.shape[0] == len(cam)
    So, go = _schur_of_oracle(Ho, bo, cam, lm3, 1e-5)
    mask = np.ones(len(cam), bool); mask[:6] = False                            # pose 0: the 1e14 prior block (see test_gpu_gtsam)
    sub = np.ix_(mask, mask)
    np.testing.assert_allclose(Sg[sub], So[sub], rtol=0, atol=1e-10 * np.abs(So[sub]).max())
    np.testing.assert_allclose(Sg, So, rtol=0, atol=1e-12 * np.abs(So).max())
    np.testing.assert_allclose(gg[mask], go[mask], rtol=0, atol=1e-10 * np.abs(go[mask]).max())
    _, V, Vo = _check_lm(gr, po, K, iters=15)
    assert np.abs(V[base:, :3] - Vo[base:, :3]).max() < 1e-6                   # landmarks
    assert np.abs(V[K:3 * K, :6] - Vo[K:3 * K, :6]).max() < 1e-6               # velocities, biases


@pytest.mark.parametrize("incremental", ["1", "0"])
def test_isam2_update_after_batch_lm_on_a_ba_graph(monkeypatch, incremental):
    """fgo_optimize_gtsam builds the structure with the landmarks eliminated; a following
    fgo_isam2_update must not run on that structure (k_ba_linearize without its buffers).  It switches the context to the
    generic form; the step equals the one of a context that never eliminated anything."""
    p = S.ba_problem(60, 1300)
    monkeypatch.setenv("FGO_ISAM_INCREMENTAL", incremental)
    monkeypatch.setenv("FGO_BA_SCHUR", "1")
    g1 = S.ba_graph(p)
    g1.optimize_gtsam(2)
    e1 = g1.error()
    st1 = g1.isam2_update(0.01)
    monkeypatch.setenv("FGO_BA_SCHUR", "0")
    g0 = S.ba_graph(p)
    g0.optimize_gtsam(2)
    st0 = g0.isam2_update(0.01)
    assert st1.structure_rebuilt == 1
    assert abs(st1.chi2_final - st0.chi2_final) <= 1e-9 * st0.chi2_final
    assert 0.5 * st1.chi2_final <= e1 * (1 + 1e-9)
    assert np.abs(g1.get_poses() - g0.get_poses()).max() < 1e-8
    st1b = g1.isam2_update(0.01)                                               # and again, on the structure it now owns
    st0b = g0.isam2_update(0.01)
    assert abs(st1b.chi2_final - st0b.chi2_final) <= 1e-9 * st0b.chi2_final


def test_linearize_reports_the_same_n_free_with_and_without_dense_output():
    """query n_free, allocate (6 n)^2, call again -- the second call (generic form: the dense system
    covers the landmarks) must not write more than the first one announced"""
    p = S.ba_problem(50, 1000)
    gr = S.ba_graph(p)
    nf = G.C.c_int64(); chi = G.C.c_double()
    gr._chk(G.lib.fgo_linearize(gr._h, G.C.byref(chi), None, None, G.C.byref(nf)))
    assert nf.value == 50 + 1000 == gr.stats().n_free
    m = 6 * nf.value
    guard = 1024
    H = np.full(m * m + guard, -7.0); b = np.full(m + guard, -7.0)
    nf2 = G.C.c_int64()
    gr._chk(G.lib.fgo_linearize(gr._h, G.C.byref(chi), S._dp(H), S._dp(b), G.C.byref(nf2)))
    assert nf2.value == nf.value
    assert np.all(H[m * m:] == -7.0) and np.all(b[m:] == -7.0)
    Hd = H[:m * m].reshape(m, m)
    assert np.allclose(Hd, Hd.T) and np.all(np.diag(Hd) > 0)
