"""CPU tests of the oracle's IMU restatement (on-manifold preintegration + CombinedImuFactor; parity unpinned, see
oracle/orc_imu.h).  Pinned by derivation: the preintegrated deltas against an independent fine-step numerical
integration, the bias Jacobians against re-integration, the factor Jacobians against central differences, and the
covariance's basic laws."""
import numpy as np
import pytest

from tests import orc_binding as orc
from tests.util import random_pose, quat_mul, quat_rot


def imu_samples(rng, n, scale_w=0.4, scale_a=1.5):
    """smooth body-frame angular rate and specific force, 200 Hz (test_vro_imu_graph.cpp:111: dt = 0.005)"""
    t = np.arange(n) * 0.005
    w = np.stack([scale_w * np.sin(2.1 * t + p) for p in rng.uniform(0, 6, 3)], 1)
    a = np.stack([scale_a * np.cos(1.3 * t + p) for p in rng.uniform(0, 6, 3)], 1) + np.array([0, 0, -9.71])
    return a, w


def test_preintegration_matches_independent_integration():
    rng = np.random.default_rng(0)
    acc, gyro = imu_samples(rng, 40)
    bhat = np.array([0.02, -0.01, 0.03, 0.002, -0.001, 0.0015])
    pim = orc.Preint(bhat, acc, gyro, 0.005)
    assert pim.dt == pytest.approx(0.2)
    # independent restatement of the same discrete update in numpy (quaternion composition, Euler on p and v)
    q = np.array([0, 0, 0, 1.0]); p = np.zeros(3); v = np.zeros(3)
    for a, w in zip(acc, gyro):
        a = a - bhat[:3]; w = w - bhat[3:]
        Ra = quat_rot(q, a)
        p = p + v * 0.005 + 0.5 * Ra * 0.005 ** 2
        v = v + Ra * 0.005
        th = np.linalg.norm(w) * 0.005
        dq = np.concatenate([np.sin(th / 2) * w / np.linalg.norm(w), [np.cos(th / 2)]])
        q = quat_mul(q, dq); q /= np.linalg.norm(q)
    np.testing.assert_allclose(pim.dp, p, atol=1e-14)
    np.testing.assert_allclose(pim.dv, v, atol=1e-14)
    np.testing.assert_allclose(pim.dR * np.sign(pim.dR[3]), q * np.sign(q[3]), atol=1e-14)


def test_bias_jacobians_vs_reintegration():
    rng = np.random.default_rng(1)
    acc, gyro = imu_samples(rng, 40)
    bhat = np.zeros(6)
    pim = orc.Preint(bhat, acc, gyro, 0.005)
    h = 1e-6
    for k in range(6):
        d = np.zeros(6); d[k] = h
        plus, minus = orc.Preint(bhat + d, acc, gyro, 0.005), orc.Preint(bhat - d, acc, gyro, 0.005)
        ddp = (plus.dp - minus.dp) / (2 * h); ddv = (plus.dv - minus.dv) / (2 * h)
        # rotation: log(dR^-1 dR(b + d)) / h
        qinv = pim.dR * np.array([-1, -1, -1, 1])
        dth = (orc.logmap(np.concatenate([[0, 0, 0], quat_mul(qinv, plus.dR)]))[:3]
               - orc.logmap(np.concatenate([[0, 0, 0], quat_mul(qinv, minus.dR)]))[:3]) / (2 * h)
        if k < 3:
            np.testing.assert_allclose(pim.J_p_ba[:, k], ddp, atol=1e-7)
            np.testing.assert_allclose(pim.J_v_ba[:, k], ddv, atol=1e-7)
            np.testing.assert_allclose(dth, 0, atol=1e-9)
        else:
            np.testing.assert_allclose(pim.J_p_bg[:, k - 3], ddp, atol=1e-7)
            np.testing.assert_allclose(pim.J_v_bg[:, k - 3], ddv, atol=1e-7)
            np.testing.assert_allclose(pim.J_R_bg[:, k - 3], dth, atol=1e-7)


def test_residual_vanishes_at_prediction_and_bias_part():
    rng = np.random.default_rng(2)
    acc, gyro = imu_samples(rng, 40)
    bhat = rng.normal(size=6) * 0.01
    pim = orc.Preint(bhat, acc, gyro, 0.005)
    xi = random_pose(rng); vi = rng.normal(size=3); bi = bhat + rng.normal(size=6) * 1e-3
    xj, vj = pim.predict(xi, vi, bi)
    r = pim.factor(xi, vi, xj, vj, bi, bi + np.array([1, 2, 3, 4, 5, 6]) * 1e-3, jac=False)
    np.testing.assert_allclose(r[:9], 0, atol=1e-12)
    np.testing.assert_allclose(r[9:], -np.array([1, 2, 3, 4, 5, 6]) * 1e-3, atol=1e-15)      # b_i - b_j


@pytest.mark.parametrize("seed", range(4))
def test_factor_jacobians_vs_central_differences(seed):
    rng = np.random.default_rng(10 + seed)
    acc, gyro = imu_samples(rng, 40)
    bhat = rng.normal(size=6) * 0.01
    pim = orc.Preint(bhat, acc, gyro, 0.005)
    xi = random_pose(rng); vi = rng.normal(size=3); bi = bhat + rng.normal(size=6) * 5e-3
    xj, vj = pim.predict(xi, vi, bi)
    if seed:                                    # away from the zero-residual point as well
        xj = orc.retract(xj, rng.normal(size=6) * 0.05); vj = vj + rng.normal(size=3) * 0.05
    bj = bi + rng.normal(size=6) * 1e-3
    r, (Jxi, Jvi, Jxj, Jvj, Jbi, Jbj) = pim.factor(xi, vi, xj, vj, bi, bj)
    h = 1e-6

    def num(which, dim):
        N = np.zeros((15, dim))
        for k in range(dim):
            d = np.zeros(dim); d[k] = h
            args_p = [xi, vi, xj, vj, bi, bj]; args_m = list(args_p)
            if which in (0, 2):
                args_p[which] = orc.retract(args_p[which], d); args_m[which] = orc.retract(args_m[which], -d)
            else:
                args_p[which] = args_p[which] + d; args_m[which] = args_m[which] - d
            N[:, k] = (pim.factor(*args_p, jac=False) - pim.factor(*args_m, jac=False)) / (2 * h)
        return N
    for which, (J, dim) in enumerate([(Jxi, 6), (Jvi, 3), (Jxj, 6), (Jvj, 3), (Jbi, 6), (Jbj, 6)]):
        np.testing.assert_allclose(J, num(which, dim), atol=2e-7, err_msg="block %d" % which)


def test_vio_graph_assembly_and_lm():
    """IMU factors inside a problem: H/b against a dense J^T W J assembled in numpy from the factor functions; GTSAM LM
    recovers the accelerometer/gyro bias the measurements were generated with (preintegration used bias 0)."""
    from tests.util import vio_graph, mixed_oracle, info_full
    rng = np.random.default_rng(4)
    g = vio_graph(rng, n_kf=6, with_planes=False)
    p = mixed_oracle(g)
    H, b = p.dense_system()
    N = len(g["values"]); m = 6 * N
    Href = np.zeros((m, m)); bref = np.zeros(m)
    v = g["values"]
    def add(J, W, r):
        nonlocal Href, bref
        Href += J.T @ W @ J; bref -= J.T @ W @ r
    for k in range(len(g["ei"])):                       # between factors
        i, j = g["ei"][k], g["ej"][k]
        e, Ji, Jj = orc.between(v[i], v[j], g["meas"][k]); J = np.zeros((6, m))
        J[:, 6 * i:6 * i + 6] = Ji; J[:, 6 * j:6 * j + 6] = Jj
        add(J, info_full(g["info"][k]), e)
    for q, vid in enumerate(g["prior_ids"]):
        W = info_full(g["prior_info"][q]); J = np.zeros((6, m))
        if g["vkind"][vid] == 0:
            e, Jp = orc.prior(v[vid], g["prior_mean"][q]); J[:, 6 * vid:6 * vid + 6] = Jp
        else:
            dim = 6 if g["vkind"][vid] == orc.VK_BIAS else 3
            e = np.zeros(6); e[:dim] = v[vid, :dim] - g["prior_mean"][q][:dim]
            J[:dim, 6 * vid:6 * vid + dim] = np.eye(dim)
        add(J, W, e)
    for f, ids in enumerate(g["imu_ids"]):
        r, Js = g["imu_pre"][f].factor(v[ids[0]], v[ids[1], :3], v[ids[2]], v[ids[3], :3], v[ids[4], :6], v[ids[5], :6])
        J = np.zeros((15, m))
        for u, (vid, Ju) in enumerate(zip(ids, Js)):
            J[:, 6 * vid:6 * vid + Ju.shape[1]] = Ju
        add(J, g["imu_info"][f], r)
    for vid in range(N):                                # padding of the 3-dof variables
        if g["vkind"][vid] == orc.VK_VEC3:
            for r_ in range(3, 6):
                Href[6 * vid + r_, 6 * vid + r_] += 1
    np.testing.assert_allclose(H, Href, rtol=0, atol=1e-12 * np.abs(Href).max())
    mask = np.ones(m, bool); mask[:6] = False
    np.testing.assert_allclose(b[mask], bref[mask], rtol=0, atol=1e-9 * np.abs(bref[mask]).max())
    e0 = p.error_gtsam()
    rc, st = p.optimize_gtsam(100)
    assert p.error_gtsam() < 1e-2 * e0
    K = g["n_kf"]
    est = p.get_poses()
    assert np.abs(est[:K, :3] - g["truth_X"][:, :3]).max() < 0.05          # poses pulled back to the truth
    assert np.abs(est[K:2 * K, :3] - g["truth_V"]).max() < 0.2


def test_covariance_laws():
    rng = np.random.default_rng(3)
    acc, gyro = imu_samples(rng, 40)
    pim = orc.Preint(np.zeros(6), acc, gyro, 0.005)
    C = pim.cov
    np.testing.assert_allclose(C, C.T, atol=1e-18)
    assert np.linalg.eigvalsh(C).min() > 0
    # bias random walk: variance grows linearly with time, sigma^2 * t (isotropic, no mixing into itself)
    d2r = np.pi / 180
    ba = ((0.04e-3 * 9.81) * np.sqrt(200)) ** 2; bg = ((10 * d2r / 3600) * np.sqrt(200)) ** 2
    np.testing.assert_allclose(np.diag(C)[9:12], ba * 0.2, rtol=1e-12)
    np.testing.assert_allclose(np.diag(C)[12:15], bg * 0.2, rtol=1e-12)
    # doubling the window roughly doubles the rotation variance
    acc2, gyro2 = imu_samples(np.random.default_rng(3), 80)
    C2 = orc.Preint(np.zeros(6), acc2, gyro2, 0.005).cov
    assert 1.5 < np.trace(C2[:3, :3]) / np.trace(C[:3, :3]) < 3.0
