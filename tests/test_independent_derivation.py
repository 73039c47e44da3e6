"""CPU: the oracle's EdgeSE3 / VertexSE3 / Levenberg restatement (oracle/orc_se3.h, orc_graph.c) against tests/se3_independent.py -- a second
evaluation written from SURVEY.md Appendix A.1's prose with 4x4 matrices and automatic differentiation, sharing no formula with the oracle
or the product.  It does not pin parity to g2o (nothing offline can); it removes "one author, one algebra"."""
import numpy as np
import pytest

from tests import orc_binding as orc
from tests import se3_independent as ind
from tests.util import random_pose, random_info, info_ut, pose_mul, pose_inv, noisy, quat_mul


def triples(rng, n, near=False):
    """(xi, xj, z): arbitrary relative poses -- the residual rotation covers the whole sphere, both signs of w_e -- or (near) z close to
    Xi^-1 Xj as in an optimisation"""
    out = []
    for _ in range(n):
        xi, xj = random_pose(rng, 3.0), random_pose(rng, 3.0)
        z = noisy(rng, pose_mul(pose_inv(xi), xj), 0.05, 0.03) if near else random_pose(rng, 3.0)
        if rng.random() < 0.5:
            z[3:] *= -1.0                                  # the sign of a stored quaternion is arbitrary
        out.append((xi, xj, z))
    return out


def test_edge_error_and_jacobians_1000_random_triples_vs_automatic_differentiation():
    rng = np.random.default_rng(2026)
    neg = 0
    for xi, xj, z in triples(rng, 700) + triples(rng, 300, near=True):
        e, Ji, Jj = orc.edge_se3(xi, xj, z)
        e2, Ji2, Jj2 = ind.edge_se3_ad(xi, xj, z)
        np.testing.assert_allclose(e, e2, atol=1e-12)
        np.testing.assert_allclose(Ji, Ji2, atol=2e-11)
        np.testing.assert_allclose(Jj, Jj2, atol=2e-11)
        # raw sign of the residual quaternion q(Z^-1) q(Xi)^-1 q(Xj) as stored: both signs must occur (toVectorMQT's w >= 0 rule is exercised)
        qe = quat_mul(pose_inv(z)[3:], quat_mul(pose_inv(xi)[3:], xj[3:]))
        neg += qe[3] < 0
    assert 200 < neg < 800, neg


def test_oplus_including_compact_quaternions_longer_than_one():
    rng = np.random.default_rng(7)
    big = 0
    for k in range(300):
        x = random_pose(rng, 2.0)
        d = np.concatenate([rng.normal(size=3), rng.normal(size=3) * (0.2 if k % 3 else 0.8)])
        big += d[3:] @ d[3:] > 1
        a, b = orc.oplus(x, d), ind.oplus(x, d)
        if a[3:] @ b[3:] < 0:
            b[3:] *= -1
        np.testing.assert_allclose(a, b, atol=1e-13)
    assert big >= 20, big                                   # fromVectorMQT's identity-rotation branch was taken


def three_pose_case(seed, noise=0.08):
    """poses 0 (fixed), 1, 2; edges (0,1), (1,2), (0,2) with inconsistent noise: the smallest graph with a loop"""
    rng = np.random.default_rng(seed)
    truth = [np.array([0, 0, 0, 0, 0, 0, 1.0]), random_pose(rng, 1.5), random_pose(rng, 1.5)]
    ei, ej = np.array([0, 1, 0], np.int32), np.array([1, 2, 2], np.int32)
    meas = np.array([noisy(rng, pose_mul(pose_inv(truth[a]), truth[b]), noise, noise * 0.5) for a, b in zip(ei, ej)])
    info = np.array([info_ut(random_info(rng)) for _ in ei])
    poses = np.array([truth[0], meas[0], pose_mul(meas[0], meas[1])])   # odometry chaining (g2o_graph.cpp:118)
    fixed = np.array([1, 0, 0], np.uint8)
    return dict(poses=poses, fixed=fixed, ei=ei, ej=ej, meas=meas, info=info)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_lm_constants_three_pose_case_vs_prose_restatement(seed):
    """tau = 1e-5, nu = 2, the rho rule, the 1/3 .. 2/3 window, the 1e-3 in rho's denominator: chi2 / lambda after every iteration and the number
    of trials of one optimize(6) call equal those of the dense restatement written from the prose"""
    g = three_pose_case(seed)
    p = orc.Problem(**g)
    rc, st = p.optimize(6)
    chis, lams = p.trace()
    poses, trace, trials = ind.lm_optimize(g["poses"], g["fixed"], g["ei"], g["ej"], g["meas"], g["info"], rc)
    assert st.trials == trials
    for k in range(rc):
        assert abs(chis[k] - trace[k][0]) <= 1e-9 * max(1.0, trace[k][0]), (k, chis[k], trace[k][0])
        assert abs(lams[k] - trace[k][1]) <= 1e-8 * trace[k][1], (k, lams[k], trace[k][1])
    got = p.get_poses()
    for v in range(3):
        s = np.sign(got[v, 3:] @ poses[v, 3:])
        np.testing.assert_allclose(got[v, :3], poses[v, :3], atol=1e-9)
        np.testing.assert_allclose(got[v, 3:], s * poses[v, 3:], atol=1e-9)
