"""GPU: the product's EdgeSE3 linearisation, oplus and Levenberg controller (csrc/se3_device.hpp, kernels.hip, fgo_lm.cpp) read back through the
C-ABI and held to tests/se3_independent.py -- matrices + automatic differentiation from SURVEY A.1's prose, no formula shared with the product
or the oracle (g2o/g2o_graph.cpp:88,115-132,244-250)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from tests import se3_independent as ind
from tests.util import random_info, info_ut, quat_mul, pose_inv
from tests.test_independent_derivation import triples, three_pose_case


def test_edge_blocks_1000_random_triples_vs_automatic_differentiation():
    """every triple is a pair of free vertices joined by one edge: the device's H blocks and gradient pieces of the pair against
    J^T Omega J / -J^T Omega e with J from automatic differentiation (batches of 100 pairs through the dense read-back)"""
    rng = np.random.default_rng(4711)
    tr = triples(rng, 700) + triples(rng, 300, near=True)
    neg = 0
    for b0 in range(0, len(tr), 100):
        batch = tr[b0:b0 + 100]
        n = 2 * len(batch)
        poses = np.array([p for xi, xj, _ in batch for p in (xi, xj)])
        ei = np.arange(0, n, 2, dtype=np.int64); ej = ei + 1
        meas = np.array([z for _, _, z in batch])
        info = np.array([info_ut(random_info(rng)) for _ in batch])
        gr = G.Graph()
        gr.add_poses(poses, np.zeros(n, np.uint8))
        gr.add_edges(ei, ej, meas, info)
        chi, H, b = gr.linearize(dense=True)
        # column of every vertex in the dense system: the read-back is in elimination order
        # (fgo_linearize documents: free poses in the order of fgo_get_order)
        order = gr.get_order() if hasattr(gr, "get_order") else None
        chi_ref = 0.0
        for k, (xi, xj, z) in enumerate(batch):
            e, Ji, Jj = ind.edge_se3_ad(xi, xj, z)
            W = ind.info_full(info[k])
            chi_ref += e @ W @ e
            J = np.hstack([Ji, Jj])
            Hk, bk = J.T @ W @ J, -J.T @ W @ e
            ci, cj = (2 * k, 2 * k + 1) if order is None else (order[2 * k], order[2 * k + 1])
            idx = np.r_[6 * ci:6 * ci + 6, 6 * cj:6 * cj + 6]
            scale = np.abs(Hk).max()
            np.testing.assert_allclose(H[np.ix_(idx, idx)], Hk, atol=1e-11 * scale)
            np.testing.assert_allclose(b[idx], bk, atol=1e-11 * max(1.0, np.abs(bk).max()))
            neg += quat_mul(pose_inv(z)[3:], quat_mul(pose_inv(xi)[3:], xj[3:]))[3] < 0
        assert abs(chi - chi_ref) <= 1e-12 * chi_ref
        gr.close()
    assert 200 < neg < 800, neg


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_lm_constants_three_pose_case_vs_prose_restatement(seed):
    g = three_pose_case(seed)
    gr = G.Graph()
    gr.add_poses(g["poses"], g["fixed"])
    gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    rc, st = gr.optimize(6)
    chis, lams = gr.trace()
    poses, trace, trials = ind.lm_optimize(g["poses"], g["fixed"], g["ei"], g["ej"], g["meas"], g["info"], rc)
    assert st.trials == trials
    for k in range(rc):
        assert abs(chis[k] - trace[k][0]) <= 1e-9 * max(1.0, trace[k][0]), (k, chis[k], trace[k][0])
        assert abs(lams[k] - trace[k][1]) <= 1e-8 * trace[k][1], (k, lams[k], trace[k][1])
    got = gr.get_poses()
    for v in range(3):
        s = np.sign(got[v, 3:] @ poses[v, 3:])
        np.testing.assert_allclose(got[v, :3], poses[v, :3], atol=1e-9)
        np.testing.assert_allclose(got[v, 3:], s * poses[v, 3:], atol=1e-9)
    gr.close()


def test_oplus_with_a_compact_quaternion_longer_than_one_through_the_device_update():
    """a star of vertices that start 120-170 degrees off around a fixed centre: the first LM trial asks for |dq|^2 > 1 on some of them
    (tests/test_gpu_branches.py asserts the branch); here the whole optimize(3) call -- every accepted / rejected trial's oplus -- is held to the
    matrix restatement"""
    rng = np.random.default_rng(99)
    n = 7
    truth = [np.array([0, 0, 0, 0, 0, 0, 1.0])]
    for k in range(1, n):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        truth.append(np.concatenate([rng.normal(size=3), ax * np.sin(0.2), [np.cos(0.2)]]))
    truth = np.array(truth)
    ei = np.zeros(n - 1, np.int64); ej = np.arange(1, n, dtype=np.int64)
    from tests.util import pose_mul
    meas = np.array([pose_mul(pose_inv(truth[0]), truth[j]) for j in ej])
    info = np.array([info_ut(np.diag([100.0] * 3 + [400.0] * 3)) for _ in ej])
    start = truth.copy()
    for j in range(1, n):                                    # rotate every leaf far away from its measurement
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = np.deg2rad(rng.uniform(120, 170))
        start[j, 3:] = quat_mul(start[j, 3:], np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]]))
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    gr = G.Graph()
    gr.add_poses(start, fixed)
    gr.add_edges(ei, ej, meas, info)
    rc, st = gr.optimize(3)
    chis, lams = gr.trace()
    poses, trace, trials = ind.lm_optimize(start, fixed, ei, ej, meas, info, rc)
    assert st.trials == trials
    for k in range(rc):
        assert abs(chis[k] - trace[k][0]) <= 1e-9 * max(1.0, trace[k][0]), (k, chis[k], trace[k][0])
    gr.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_device_lm_converges_to_the_minimum_scipy_finds(seed):
    """tests/test_converged_optimum.py on the device: the minimum of the objective as SURVEY A.1's prose defines it, found by
    scipy.optimize.least_squares with a finite-difference Jacobian, against where the device's LM (10 x optimize(2), g2o/g2o_graph.cpp:244-250,
    and five more calls) comes to rest: final chi2 1e-8 relative (north_star: 1e-6), poses 1e-6"""
    from tests.test_converged_optimum import case, scipy_optimum, compare_poses
    g = case(seed)
    chi_ref, poses_ref = scipy_optimum(g)
    gr = G.Graph()
    gr.add_poses(g["poses"], g["fixed"])
    gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    for _ in range(15):
        gr.optimize(2)
    chi = gr.chi2()
    assert abs(chi - chi_ref) <= 1e-8 * chi_ref, (chi, chi_ref)
    compare_poses(gr.get_poses(), poses_ref, 1e-6)
    gr.close()


def test_gtsam_between_and_prior_blocks_vs_matrix_logarithm():
    """the GTSAM-semantics twin: BetweenFactor<Pose3> / PriorFactor<Pose3> as the device linearises them (kernels_gtsam.hip, pose3_device.hpp) against
    tests/pose3_independent.py -- matrix logarithm / exponential at 40 digits, Jacobians by numerical differentiation of the prose definitions"""
    from tests import pose3_independent as p3
    from tests.test_independent_pose3 import _triples
    rng = np.random.default_rng(4712)
    tr = _triples(rng, 24)
    n = 2 * len(tr)
    poses = np.array([p for xi, xj, _ in tr for p in (xi, xj)])
    ei = np.arange(0, n, 2, dtype=np.int64); ej = ei + 1
    meas = np.array([z for _, _, z in tr])
    info = np.array([info_ut(random_info(rng)) for _ in tr])
    prior_info = info_ut(random_info(rng))
    gr = G.Graph()
    gr.add_poses(poses, np.zeros(n, np.uint8))
    gr.add_edges(ei, ej, meas, info, tangent_order=G.FGO_TANGENT_GTSAM)
    pm = poses[0].copy(); pm[:3] += 0.1                          # a prior on vertex 0 whose mean is NOT its value
    gr.add_prior(0, pm, prior_info)
    chi, H, b = gr.linearize(dense=True)
    for k, (xi, xj, z) in enumerate(tr):
        e, Ji, Jj = p3.between(xi, xj, z)
        W = ind.info_full(info[k])
        J = np.hstack([Ji, Jj])
        Hk, bk = J.T @ W @ J, -J.T @ W @ e
        if k == 0:
            ep, Jp = p3.prior(xi, pm)
            Wp = ind.info_full(prior_info)
            Hk[:6, :6] += Jp.T @ Wp @ Jp; bk[:6] += -Jp.T @ Wp @ ep
        idx = np.r_[12 * k:12 * k + 12]
        np.testing.assert_allclose(H[np.ix_(idx, idx)], Hk, atol=1e-9 * np.abs(Hk).max())
        np.testing.assert_allclose(b[idx], bk, atol=1e-9 * max(1.0, np.abs(bk).max()))
    gr.close()


def test_reprojection_blocks_vs_automatic_differentiation():
    """GenericProjectionFactor as the device linearises it (generic form, kernels_gtsam.hip / factors_device.hpp) against tests/camera_independent.py:
    every pair (pose k, point k) carries one observation; H_xx, H_xp, H_pp and both gradient pieces, incl. points behind the camera"""
    from tests import camera_independent as cam
    from tests.util import random_pose, SR4000_CALIB, pose_mul, quat_rot
    rng = np.random.default_rng(2028)
    n = 60
    bps = random_pose(rng, 0.1)
    poses = np.array([random_pose(rng, 1.0) for _ in range(n)])
    pts, uvs = [], []
    for k in range(n):
        c = pose_mul(poses[k], bps)
        local = np.array([rng.normal() * 0.4, rng.normal() * 0.4, rng.uniform(1, 6) * (-1 if k % 12 == 0 else 1)])
        pts.append(c[:3] + quat_rot(c[3:], local)); uvs.append(rng.uniform(0, 180, size=2))
    sigma = 0.7
    gr = G.Graph()
    gr.add_poses(poses, np.zeros(n, np.uint8))
    gr.set_calibration(SR4000_CALIB, bps)
    for k in range(n):
        gr.add_point(n + k, pts[k])
        gr.add_reproj(k, n + k, uvs[k], sigma)
    chi, H, b = gr.linearize(dense=True)
    assert H.shape == (12 * n, 12 * n)
    chi_ref = 0.0
    for k in range(n):
        r, Hx, Hp = cam.reproj_ad(poses[k], pts[k], uvs[k], SR4000_CALIB, bps)
        w = 1.0 / sigma ** 2
        chi_ref += w * r @ r
        ix, ip = np.r_[6 * k:6 * k + 6], np.r_[6 * (n + k):6 * (n + k) + 3]
        sc = max(1.0, w * np.abs(Hx).max() ** 2)
        np.testing.assert_allclose(H[np.ix_(ix, ix)], w * Hx.T @ Hx, atol=1e-9 * sc)
        np.testing.assert_allclose(H[np.ix_(ix, ip)], w * Hx.T @ Hp, atol=1e-9 * sc)
        np.testing.assert_allclose(H[np.ix_(ip, ip)], w * Hp.T @ Hp, atol=1e-9 * sc)
        np.testing.assert_allclose(b[ix], -w * Hx.T @ r, atol=1e-9 * sc)
        np.testing.assert_allclose(b[ip], -w * Hp.T @ r, atol=1e-9 * sc)
    assert abs(chi - chi_ref) <= 1e-11 * chi_ref
    gr.close()


def test_combined_imu_factor_blocks_vs_matrix_logarithm():
    """CombinedImuFactor as the device linearises it (k_imu_eval + k_imu_blocks: [J r]^T W [J r] on f64 MFMA, straight into H) against
    tests/imu_independent.py: a graph of nothing but one factor per key-frame pair (X_i, V_i, X_j, V_j, B_i, B_j); the 15 x 15 weight is the one
    the product derives from the payload (its inverse-covariance is tested separately, tests/test_gpu_imu.py)"""
    from tests import imu_independent as imu
    from tests import orc_binding as orc
    from tests.test_independent_imu import _case
    rng = np.random.default_rng(315)
    for _ in range(3):
        xi, vi, xj, vj, bi, bj, pim = _case(rng)
        gr = G.Graph()
        gr.add_poses(np.array([xi, xj]))
        gr.add_vec3(2, vi); gr.add_vec3(3, vj)
        gr.add_bias(4, bi); gr.add_bias(5, bj)
        gr.set_gravity(orc.GRAVITY)
        gr.add_imu([0, 2, 1, 3, 4, 5], pim.buf)              # X(i) V(i) X(j) V(j) B(i) B(j): test_ba_imu_graph.cpp:239-241
        chi, H, b = gr.linearize(dense=True)
        assert H.shape == (36, 36)
        r, Js = imu.factor(xi, vi, xj, vj, bi, bj, pim, orc.GRAVITY)
        W = G.preint_information(pim.buf)
        # dense order = order added: X_i X_j V_i V_j B_i B_j, six scalars each (the 3-dof velocities are padded)
        J = np.zeros((15, 36))
        for name, Jk in zip(("xi", "vi", "xj", "vj", "bi", "bj"), Js):
            c0 = {"xi": 0, "xj": 6, "vi": 12, "vj": 18, "bi": 24, "bj": 30}[name]
            J[:, c0:c0 + Jk.shape[1]] = Jk
        Href, bref = J.T @ W @ J, -J.T @ W @ r
        pad = np.zeros(36, bool); pad[15:18] = True; pad[21:24] = True            # the padding rows / columns of the two velocities
        scale = np.abs(Href).max()
        np.testing.assert_allclose(H[np.ix_(~pad, ~pad)], Href[np.ix_(~pad, ~pad)], atol=1e-8 * scale)
        np.testing.assert_allclose(b[~pad], bref[~pad], atol=1e-8 * max(1.0, np.abs(bref).max()))
        assert abs(chi - r @ W @ r) <= 1e-9 * (r @ W @ r)
        gr.close()
