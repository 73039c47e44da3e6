"""CPU: oracle/orc_imu.h (CombinedImuFactor residual + closed-form Jacobians w.r.t. X_i, V_i, X_j, V_j, B_i, B_j) against tests/imu_independent.py
(matrix exponential / logarithm at 40 digits, Jacobians by numerical differentiation of the definition)."""
import numpy as np

from tests import orc_binding as orc
from tests import imu_independent as imu
from tests.util import random_pose, noisy


def _case(rng):
    samples = 40
    t = np.arange(samples) * 0.005
    gyro = np.stack([0.6 * np.sin(2.1 * t + p) for p in rng.uniform(0, 6, 3)], 1)
    acc = np.stack([1.5 * np.cos(1.3 * t + p) for p in rng.uniform(0, 6, 3)], 1) + np.array([0, 0, -9.71])
    bhat = rng.normal(size=6) * 0.01
    pim = orc.Preint(bhat, acc, gyro, 0.005)
    xi = random_pose(rng, 1.0); vi = rng.normal(size=3)
    bi = bhat + rng.normal(size=6) * 0.02                              # the bias moved away from the integration bias: the correction terms matter
    xj_pred, vj_pred = pim.predict(xi, vi, bi)
    xj = noisy(rng, xj_pred, 0.1, 0.15); vj = vj_pred + rng.normal(size=3) * 0.2
    bj = bi + rng.normal(size=6) * 0.01
    return xi, vi, xj, vj, bi, bj, pim


def test_combined_imu_factor_residual_and_jacobians_vs_matrix_logarithm():
    rng = np.random.default_rng(314)
    g = orc.GRAVITY
    for _ in range(6):
        xi, vi, xj, vj, bi, bj, pim = _case(rng)
        r, Js = pim.factor(xi, vi, xj, vj, bi, bj, g=g)
        r2, Js2 = imu.factor(xi, vi, xj, vj, bi, bj, pim, g)
        np.testing.assert_allclose(r, r2, atol=1e-11)
        assert np.linalg.norm(r2[:3]) > 0.05                           # a real rotation residual, not the identity
        for J, J2 in zip(Js, Js2):
            np.testing.assert_allclose(J, J2, atol=1e-9 * max(1.0, np.abs(J2).max()))
