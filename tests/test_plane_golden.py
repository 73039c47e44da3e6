"""Golden vectors the reference itself holds for the plane path: the GTSAM unit tests copied
into gtsam/test/.  These are data (inputs + expected outputs), transcribed from
  gtsam/test/testOrientedPlane3.cpp:61-70      transform known answer (tol 1e-9)
  gtsam/test/testOrientedPlane3.cpp:72-90      transform Jacobians vs numerical derivative (tol 1e-9)
  gtsam/test/testOrientedPlane3.cpp:111-140    retract / localCoordinates round trip, 10 000 random trials (tol 1e-6)
  gtsam/test/testOrientedPlane3.cpp:143-149    errorVector regression (tol 1e-5) and its definition
  gtsam/test/testOrientedPlane3Factor.cpp:37-81    two measurements differing in range  -> d = 2.0
  gtsam/test/testOrientedPlane3Factor.cpp:84-126   two measurements differing in angle  -> n = (-√2/2, -√2/2, 0), d = 3
They pin the oracle's OrientedPlane3 restatement (oracle/orc_plane.h); the GPU plane kernel is then pinned to the
oracle in tests/test_gpu_factors.py."""
import numpy as np
import pytest

from tests import orc_binding as orc
from tests.util import info_ut


def ypr_pose(yaw, pitch, roll, t):
    """gtsam::Rot3::Ypr(y, p, r) = Rz(y) Ry(p) Rx(r) as a unit quaternion (x, y, z, w)"""
    cy, sy, cp, sp, cr, sr = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(roll / 2), np.sin(roll / 2)
    q = np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])
    return np.concatenate([t, q])


def test_transform_known_answer():
    pose = ypr_pose(-np.pi / 4, 0.0, 0.0, [2.0, 3.0, 4.0])
    plane = orc.plane(-1, 0, 0, 5)
    expected = orc.plane(-np.sqrt(2) / 2, -np.sqrt(2) / 2, 0.0, 3)
    np.testing.assert_allclose(orc.plane_transform(plane, pose), expected, atol=1e-9)


def test_transform_jacobians_vs_numerical_derivative():
    pose = ypr_pose(-np.pi / 4, 0.0, 0.0, [2.0, 3.0, 4.0])
    plane = orc.plane(-1, 0, 0, 5)
    out, Hx, Hp = orc.plane_transform(plane, pose, jac=True)
    h = 1e-6
    Nx = np.zeros((3, 6)); Np = np.zeros((3, 3))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        Nx[:, k] = (orc.plane_local(out, orc.plane_transform(plane, orc.retract(pose, d)))
                    - orc.plane_local(out, orc.plane_transform(plane, orc.retract(pose, -d)))) / (2 * h)
    for k in range(3):
        d = np.zeros(3); d[k] = h
        Np[:, k] = (orc.plane_local(out, orc.plane_transform(orc.plane_retract(plane, d), pose))
                    - orc.plane_local(out, orc.plane_transform(orc.plane_retract(plane, -d), pose))) / (2 * h)
    np.testing.assert_allclose(Hx, Nx, atol=1e-9)
    np.testing.assert_allclose(Hp, Np, atol=1e-9)
    rng = np.random.default_rng(0)                      # and at generic configurations
    for _ in range(5):
        q = rng.normal(size=4); pose = np.concatenate([rng.normal(size=3) * 3, q / np.linalg.norm(q)])
        plane = orc.plane(*rng.uniform(-1, 1, 3), rng.uniform(0.01, 10))
        out, Hx, Hp = orc.plane_transform(plane, pose, jac=True)
        for k in range(6):
            d = np.zeros(6); d[k] = h
            Nx[:, k] = (orc.plane_local(out, orc.plane_transform(plane, orc.retract(pose, d)))
                        - orc.plane_local(out, orc.plane_transform(plane, orc.retract(pose, -d)))) / (2 * h)
        for k in range(3):
            d = np.zeros(3); d[k] = h
            Np[:, k] = (orc.plane_local(out, orc.plane_transform(orc.plane_retract(plane, d), pose))
                        - orc.plane_local(out, orc.plane_transform(orc.plane_retract(plane, -d), pose))) / (2 * h)
        np.testing.assert_allclose(Hx, Nx, atol=1e-8)
        np.testing.assert_allclose(Hp, Np, atol=1e-8)


def test_retract_local_roundtrip_10000():
    rng = np.random.default_rng(1)
    for _ in range(10000):
        p1 = orc.plane(*rng.uniform(-1, 1, 3), rng.uniform(0.01, 10.0))
        v12 = np.array([rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi), rng.uniform(-10, 10)])
        if np.linalg.norm(v12) > np.pi:                  # "magnitude of the rotation can be at most pi" (:125-127)
            v12 = v12 / np.pi
        p2 = orc.plane_retract(p1, v12)
        back = orc.plane_local(p1, p2)
        np.testing.assert_allclose(back, v12, atol=1e-6)
        np.testing.assert_allclose(orc.plane_retract(p1, back), p2, atol=1e-6)


def test_error_vector_regression():
    plane1 = orc.plane(-1, 0.1, 0.2, 5)
    plane2 = orc.plane(-1.1, 0.2, 0.3, 5.4)
    np.testing.assert_allclose(orc.plane_error_vector(plane1, plane1), 0, atol=1e-8)
    np.testing.assert_allclose(orc.plane_error_vector(plane1, plane2), [-0.0677674148, -0.0760543588, -0.4], atol=1e-5)
    assert orc.plane_error_vector(plane1, plane2)[2] == pytest.approx(plane1[3] - plane2[3])


def _one_linear_step(pose, lm, measurements, sig_pose=1e-3, sig_meas=0.1):
    """what isam2.update(graph, values) + calculateEstimate() does for these tiny graphs: one Gauss-Newton step
    on {PriorFactor<Pose3>(sigma 1e-3), OrientedPlane3Factor x 2 (sigma 0.1)} from the initial values"""
    H = np.zeros((9, 9)); b = np.zeros(9)
    e, J = orc.prior(pose, pose)
    W = np.eye(6) / sig_pose ** 2
    H[:6, :6] += J.T @ W @ J; b[:6] -= J.T @ W @ e
    for z in measurements:
        r, Hx, Hp = orc.plane_factor(pose, lm, z)
        Jf = np.hstack([Hx, Hp]); Wm = np.eye(3) / sig_meas ** 2
        H += Jf.T @ Wm @ Jf; b -= Jf.T @ Wm @ r
    d = np.linalg.solve(H, b)
    return orc.retract(pose, d[:6]), orc.plane_retract(lm, d[6:])


def test_factor_fusion_range():
    pose = ypr_pose(0, 0, 0, [0.0, 0.0, 0.0])
    lm0 = orc.plane(-1.0, 0.0, 0.0, 3.0)
    _, lm = _one_linear_step(pose, lm0, [orc.plane(-1.0, 0.0, 0.0, 3.0), orc.plane(-1.0, 0.0, 0.0, 1.0)])
    np.testing.assert_allclose(lm, orc.plane(-1.0, 0.0, 0.0, 2.0), atol=1e-9)


def test_factor_fusion_angle():
    pose = ypr_pose(0, 0, 0, [0.0, 0.0, 0.0])
    lm0 = orc.plane(-1.0, 0.0, 0.0, 3.0)
    _, lm = _one_linear_step(pose, lm0, [orc.plane(-1.0, 0.0, 0.0, 3.0), orc.plane(0.0, -1.0, 0.0, 3.0)])
    np.testing.assert_allclose(lm, orc.plane(-np.sqrt(2) / 2, -np.sqrt(2) / 2, 0.0, 3.0), atol=1e-9)


def test_factor_jacobians_exact_at_zero_residual():
    """the GTSAM 4.0 factor takes d r / d predicted = I; that is exact where the residual vanishes"""
    rng = np.random.default_rng(2)
    q = rng.normal(size=4); pose = np.concatenate([rng.normal(size=3), q / np.linalg.norm(q)])
    pl = orc.plane(*rng.uniform(-1, 1, 3), 4.0)
    z = orc.plane_transform(pl, pose)
    r, Hx, Hp = orc.plane_factor(pose, pl, z)
    np.testing.assert_allclose(r, 0, atol=1e-12)
    h = 1e-6; Nx = np.zeros((3, 6)); Np = np.zeros((3, 3))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        Nx[:, k] = (orc.plane_factor(orc.retract(pose, d), pl, z, jac=False) - orc.plane_factor(orc.retract(pose, -d), pl, z, jac=False)) / (2 * h)
    for k in range(3):
        d = np.zeros(3); d[k] = h
        Np[:, k] = (orc.plane_factor(pose, orc.plane_retract(pl, d), z, jac=False) - orc.plane_factor(pose, orc.plane_retract(pl, -d), z, jac=False)) / (2 * h)
    np.testing.assert_allclose(Hx, Nx, atol=1e-8)
    np.testing.assert_allclose(Hp, Np, atol=1e-8)
