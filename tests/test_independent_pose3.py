"""CPU: the oracle's Pose3 restatement (oracle/orc_pose3.h: Expmap / Logmap / retract, PriorFactor and BetweenFactor residuals with their
closed-form SE(3) Jacobians) against tests/pose3_independent.py -- matrix exponential / logarithm at 40 digits and numerical differentiation of the
prose definitions, no formula shared (round 6; the GTSAM twin of tests/test_independent_derivation.py; gtsam/gtsam_graph.cpp:338-341, 689-692)."""
import numpy as np

from tests import orc_binding as orc
from tests import pose3_independent as p3
from tests.util import random_pose, pose_mul, pose_inv, noisy


def _triples(rng, n):
    out = []
    for k in range(n):
        xi, xj = random_pose(rng, 2.0), random_pose(rng, 2.0)
        # residual rotations from a few degrees (an optimisation) to ~170 degrees (Logmap far from the identity); never at pi, where the chart ends
        if k % 2:
            z = noisy(rng, pose_mul(pose_inv(xi), xj), 0.05, 0.03)
        else:
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            ang = np.deg2rad(rng.uniform(5.0, 170.0))
            off = np.concatenate([rng.normal(size=3) * 0.5, ax * np.sin(ang / 2), [np.cos(ang / 2)]])
            z = pose_mul(pose_mul(pose_inv(xi), xj), off)
        out.append((xi, xj, z))
    return out


def test_between_factor_residual_and_jacobians_vs_matrix_logarithm():
    rng = np.random.default_rng(77)
    worst = 0.0
    for xi, xj, z in _triples(rng, 40):
        e, Ji, Jj = orc.between(xi, xj, z)
        e2, Ji2, Jj2 = p3.between(xi, xj, z)
        np.testing.assert_allclose(e, e2, atol=1e-12)
        np.testing.assert_allclose(Ji, Ji2, atol=1e-10)
        np.testing.assert_allclose(Jj, Jj2, atol=1e-10)
        worst = max(worst, np.linalg.norm(e2[:3]))
    assert worst > 2.0                                      # residual rotations well away from the identity were part of it


def test_prior_factor_and_retract_vs_matrix_exponential():
    rng = np.random.default_rng(78)
    for _ in range(25):
        x = random_pose(rng, 2.0)
        m = pose_mul(x, noisy(rng, np.array([0, 0, 0, 0, 0, 0, 1.0]), 0.3, 0.3))
        e, J = orc.prior(x, m)
        e2, J2 = p3.prior(x, m)
        np.testing.assert_allclose(e, e2, atol=1e-12)
        np.testing.assert_allclose(J, J2, atol=1e-10)
        d = np.concatenate([rng.normal(size=3) * 0.7, rng.normal(size=3)])
        a, b = orc.retract(x, d), p3.retract(x, d)
        if a[3:] @ b[3:] < 0:
            b[3:] *= -1
        np.testing.assert_allclose(a, b, atol=1e-13)
