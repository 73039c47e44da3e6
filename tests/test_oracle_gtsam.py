"""CPU tests of the oracle's GTSAM-semantics restatement (parity unpinned — GTSAM 4.0 is
not vendored).  Pinned by derivation: Expmap/Logmap round trips, factor Jacobians against central differences
through the same retraction, the AdjointMap identity the reference itself relies on
(gtsam/gtsam_graph.cpp:675-676), and GTSAM's LM restated independently in numpy."""
import numpy as np
import pytest

from tests import orc_binding as orc
from tests.util import random_pose, small_graph, info_full, info_ut, pose_mul, pose_inv


def test_expmap_logmap_roundtrip():
    rng = np.random.default_rng(0)
    for scale in (1e-12, 1e-6, 0.1, 1.0, 2.5):
        for _ in range(10):
            xi = rng.normal(size=6) * scale
            if np.linalg.norm(xi[:3]) > 3.0:
                xi[:3] *= 3.0 / np.linalg.norm(xi[:3])
            T = orc.expmap(xi)
            np.testing.assert_allclose(orc.logmap(T), xi, atol=1e-9 * max(1.0, scale))
            assert abs(np.linalg.norm(T[3:]) - 1) < 1e-14


def test_expmap_matches_matrix_exponential():
    from scipy.linalg import expm
    rng = np.random.default_rng(1)
    for _ in range(5):
        xi = rng.normal(size=6)
        w, v = xi[:3], xi[3:]
        M = np.zeros((4, 4))
        M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
        M[:3, 3] = v
        E = expm(M)
        T = orc.expmap(xi)
        np.testing.assert_allclose(T[:3], E[:3, 3], atol=1e-12)
        x, y, z, ww = T[3:]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                      [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                      [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(R, E[:3, :3], atol=1e-12)


def test_retract_local_consistency():
    rng = np.random.default_rng(2)
    x = random_pose(rng)
    xi = rng.normal(size=6) * 0.3
    y = orc.retract(x, xi)
    np.testing.assert_allclose(orc.logmap(pose_mul(pose_inv(x), y)), xi, atol=1e-12)


@pytest.mark.parametrize("seed", range(6))
def test_between_jacobians_vs_central_differences(seed):
    rng = np.random.default_rng(10 + seed)
    xi, xj, z = random_pose(rng, 2.0), random_pose(rng, 2.0), random_pose(rng, 2.0)
    if seed == 0:      # small-residual regime (the usual one) as well as the generic one
        z = pose_mul(pose_mul(pose_inv(xi), xj), orc.expmap(rng.normal(size=6) * 1e-3))
    e, Ji, Jj = orc.between(xi, xj, z)
    h = 1e-6
    Ni = np.zeros((6, 6)); Nj = np.zeros((6, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        Ni[:, k] = (orc.between(orc.retract(xi, d), xj, z, jac=False) - orc.between(orc.retract(xi, -d), xj, z, jac=False)) / (2 * h)
        Nj[:, k] = (orc.between(xi, orc.retract(xj, d), z, jac=False) - orc.between(xi, orc.retract(xj, -d), z, jac=False)) / (2 * h)
    np.testing.assert_allclose(Ji, Ni, atol=5e-8)
    np.testing.assert_allclose(Jj, Nj, atol=5e-8)


def test_prior_jacobian_and_zero_at_mean():
    rng = np.random.default_rng(20)
    m = random_pose(rng)
    np.testing.assert_allclose(orc.prior(m, m, jac=False), 0, atol=1e-15)
    x = orc.retract(m, rng.normal(size=6) * 0.4)
    e, J = orc.prior(x, m)
    N = np.zeros((6, 6)); h = 1e-6
    for k in range(6):
        d = np.zeros(6); d[k] = h
        N[:, k] = (orc.prior(orc.retract(x, d), m, jac=False) - orc.prior(orc.retract(x, -d), m, jac=False)) / (2 * h)
    np.testing.assert_allclose(J, N, atol=5e-8)


def _gtsam_graph(rng, n=12, extra=14, noise=0.03):
    g = small_graph(rng, n=n, extra=extra, noise=noise, fixed_first=False)
    prior_info = info_ut(np.diag([1e14] * 6))                      # Diagonal::Sigmas(1e-7): gtsam_graph.cpp:338-341
    return g, np.array([0], np.int32), g["poses"][:1].copy(), prior_info[None, :]


def _lm_gtsam_numpy(g, pid, pmean, pinfo, max_iters):
    poses = g["poses"].copy()
    n = len(poses); m = 6 * n

    def lin(ps):
        H = np.zeros((m, m)); b = np.zeros(m); err = 0
        for k in range(len(g["ei"])):
            i, j = g["ei"][k], g["ej"][k]
            e, Ji, Jj = orc.between(ps[i], ps[j], g["meas"][k]); W = info_full(g["info"][k]); err += 0.5 * e @ W @ e
            J = np.zeros((6, m)); J[:, 6 * i:6 * i + 6] = Ji; J[:, 6 * j:6 * j + 6] = Jj
            H += J.T @ W @ J; b -= J.T @ W @ e
        for k in range(len(pid)):
            e, Jp = orc.prior(ps[pid[k]], pmean[k]); W = info_full(pinfo[k]); err += 0.5 * e @ W @ e
            J = np.zeros((6, m)); J[:, 6 * pid[k]:6 * pid[k] + 6] = Jp
            H += J.T @ W @ J; b -= J.T @ W @ e
        return H, b, err

    lam = 1e-5; trace = []
    _, _, cur = lin(poses)
    it = 0
    while True:
        before = cur
        H, b, _ = lin(poses)
        while True:
            d = np.linalg.solve(H + lam * np.eye(m), b)
            lin_change = b @ d - 0.5 * d @ H @ d
            cand = np.array([orc.retract(poses[v], d[6 * v:6 * v + 6]) for v in range(n)])
            _, _, new = lin(cand)
            ok = stop = False
            if lin_change >= 0:
                cost = cur - new
                if lin_change > 1e-20 and cost / lin_change > 1e-3:
                    ok = True
                if abs(cost) < 1e-5 * cur:
                    stop = True
            if ok:
                poses, cur = cand, new; lam /= 10; break
            if stop:
                break
            lam *= 10
            if lam >= 1e5:
                break
        it += 1; trace.append((2 * cur, lam))
        if it >= max_iters or cur <= 0:
            break
        if (before - cur) / before <= 1e-5 or before - cur <= 1e-5:
            break
    return poses, trace


def test_gtsam_lm_matches_numpy_restatement():
    rng = np.random.default_rng(30)
    g, pid, pmean, pinfo = _gtsam_graph(rng)
    p = orc.Problem(**g); p.set_gtsam(); p.add_priors(pid, pmean, pinfo)
    rc, st = p.optimize_gtsam(100)
    poses_ref, trace = _lm_gtsam_numpy(g, pid, pmean, pinfo, 100)
    chis, lams = p.trace()
    assert rc == len(trace)
    for k in range(rc):
        assert abs(chis[k] - trace[k][0]) <= 1e-6 * max(1.0, trace[k][0])
        assert abs(lams[k] - trace[k][1]) <= 1e-9 * trace[k][1]
    np.testing.assert_allclose(p.get_poses(), poses_ref, atol=1e-7)
    assert p.error_gtsam() == pytest.approx(0.5 * chis[-1], rel=1e-9)
    assert st.chi2_final < 0.5 * st.chi2_initial
    np.testing.assert_allclose(p.get_poses()[0], g["poses"][0], atol=1e-9)     # the 1e-7-sigma prior pins pose 0
