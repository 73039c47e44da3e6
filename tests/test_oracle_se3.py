"""CPU tests of the oracle's SE3 pose-graph restatement (g2o semantics).

The reference holds no golden vectors for the g2o path (parity unpinned), so the oracle is
pinned by derivation checks: analytic Jacobians against central differences of the error definition,
H/b against a dense J^T Omega J built from those Jacobians, the sparse solve against numpy, and the LM
controller against an independent numpy restatement of the g2o update rules.
"""
import numpy as np
import pytest

from tests import orc_binding as orc
from tests.util import random_pose, small_graph, info_full, quat_mul, pose_mul, pose_inv


def test_oplus_matches_definition():
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = random_pose(rng)
        d = rng.normal(size=6) * 0.1
        y = orc.oplus(x, d)
        w = np.sqrt(1 - d[3:] @ d[3:])
        inc = np.concatenate([d[:3], d[3:], [w]])
        ref = pose_mul(x, inc)
        np.testing.assert_allclose(y, ref, atol=1e-14)
    # |dq|^2 > 1 -> identity rotation (g2o fromCompactQuaternion)
    x = random_pose(rng)
    y = orc.oplus(x, np.array([0.1, 0.2, 0.3, 0.9, 0.9, 0.9]))
    np.testing.assert_allclose(y[3:], x[3:], atol=1e-15)


def test_error_is_zero_at_measurement():
    rng = np.random.default_rng(1)
    xi, xj = random_pose(rng), random_pose(rng)
    z = pose_mul(pose_inv(xi), xj)
    e = orc.edge_se3(xi, xj, z, jac=False)
    np.testing.assert_allclose(e, 0, atol=1e-14)


def test_error_quaternion_sign_normalised():
    rng = np.random.default_rng(2)
    xi, xj, z = random_pose(rng), random_pose(rng), random_pose(rng)
    e1 = orc.edge_se3(xi, xj, z, jac=False)
    xj2 = xj.copy(); xj2[3:] *= -1          # same rotation, opposite quaternion sign
    e2 = orc.edge_se3(xi, xj2, z, jac=False)
    np.testing.assert_allclose(e1, e2, atol=1e-14)


@pytest.mark.parametrize("seed", range(8))
def test_jacobians_vs_central_differences(seed):
    rng = np.random.default_rng(100 + seed)
    xi, xj, z = random_pose(rng, 3.0), random_pose(rng, 3.0), random_pose(rng, 3.0)
    e, Ji, Jj = orc.edge_se3(xi, xj, z)
    h = 1e-6
    Ni = np.zeros((6, 6)); Nj = np.zeros((6, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        Ni[:, k] = (orc.edge_se3(orc.oplus(xi, d), xj, z, jac=False) - orc.edge_se3(orc.oplus(xi, -d), xj, z, jac=False)) / (2 * h)
        Nj[:, k] = (orc.edge_se3(xi, orc.oplus(xj, d), z, jac=False) - orc.edge_se3(xi, orc.oplus(xj, -d), z, jac=False)) / (2 * h)
    np.testing.assert_allclose(Ji, Ni, atol=2e-8)
    np.testing.assert_allclose(Jj, Nj, atol=2e-8)


def test_dense_system_is_JtWJ():
    rng = np.random.default_rng(5)
    g = small_graph(rng, n=7, extra=8)
    p = orc.Problem(**g)
    H, b = p.dense_system()
    free = [v for v in range(len(g["poses"])) if not g["fixed"][v]]
    col = {v: k for k, v in enumerate(free)}
    m = 6 * len(free)
    Href = np.zeros((m, m)); bref = np.zeros(m); chi = 0
    for k in range(len(g["ei"])):
        i, j = g["ei"][k], g["ej"][k]
        e, Ji, Jj = orc.edge_se3(g["poses"][i], g["poses"][j], g["meas"][k])
        W = info_full(g["info"][k])
        chi += e @ W @ e
        J = np.zeros((6, m))
        if i in col: J[:, 6 * col[i]:6 * col[i] + 6] = Ji
        if j in col: J[:, 6 * col[j]:6 * col[j] + 6] = Jj
        Href += J.T @ W @ J
        bref -= J.T @ W @ e
    np.testing.assert_allclose(H, Href, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(b, bref, rtol=1e-12, atol=1e-9)
    assert abs(p.chi2() - chi) <= 1e-12 * chi


def test_sparse_solve_matches_numpy():
    rng = np.random.default_rng(6)
    g = small_graph(rng, n=40, extra=60)
    p = orc.Problem(**g)
    H, b = p.dense_system()
    lam = 1e-5 * np.abs(np.diag(H)).max()
    rc, d = p.solve_step(lam)
    assert rc == 0
    ref = np.linalg.solve(H + lam * np.eye(len(b)), b)
    np.testing.assert_allclose(d, ref, rtol=1e-8, atol=1e-10)


def _lm_numpy(g, iters):
    """Independent numpy restatement of one g2o optimize(iters) call (dense solves)."""
    poses = g["poses"].copy()
    free = [v for v in range(len(poses)) if not g["fixed"][v]]
    col = {v: k for k, v in enumerate(free)}
    m = 6 * len(free)

    def lin(ps):
        H = np.zeros((m, m)); b = np.zeros(m); chi = 0
        for k in range(len(g["ei"])):
            i, j = g["ei"][k], g["ej"][k]
            e, Ji, Jj = orc.edge_se3(ps[i], ps[j], g["meas"][k])
            W = info_full(g["info"][k]); chi += e @ W @ e
            J = np.zeros((6, m))
            if i in col: J[:, 6 * col[i]:6 * col[i] + 6] = Ji
            if j in col: J[:, 6 * col[j]:6 * col[j] + 6] = Jj
            H += J.T @ W @ J; b -= J.T @ W @ e
        return H, b, chi

    def chi2(ps):
        c = 0
        for k in range(len(g["ei"])):
            e = orc.edge_se3(ps[g["ei"][k]], ps[g["ej"][k]], g["meas"][k], jac=False)
            c += e @ info_full(g["info"][k]) @ e
        return c

    trace = []
    lam, ni = 0.0, 2.0
    for it in range(iters):
        H, b, cur = lin(poses)
        if it == 0:
            lam = 1e-5 * np.abs(np.diag(H)).max(); ni = 2.0
        q = 0
        while True:
            x = np.linalg.solve(H + lam * np.eye(m), b)
            cand = poses.copy()
            for v in free:
                cand[v] = orc.oplus(poses[v], x[6 * col[v]:6 * col[v] + 6])
            tmp = chi2(cand)
            rho = (cur - tmp) / (x @ (lam * x + b) + 1e-3)
            if rho > 0 and np.isfinite(tmp):
                alpha = min(1 - (2 * rho - 1) ** 3, 2 / 3)
                lam *= max(1 / 3, alpha); ni = 2.0; cur = tmp; poses = cand
            else:
                lam *= ni; ni *= 2
            q += 1
            if not (rho < 0 and q < 10):
                break
        trace.append((cur, lam))
    return poses, trace


def test_lm_matches_numpy_restatement():
    rng = np.random.default_rng(7)
    g = small_graph(rng, n=12, extra=15, noise=0.05)
    p = orc.Problem(**g)
    rc, st = p.optimize(4)
    assert rc == 4
    chis, lams = p.trace()
    poses_ref, trace = _lm_numpy(g, 4)
    for k in range(4):
        assert abs(chis[k] - trace[k][0]) <= 1e-8 * max(1.0, trace[k][0])
        assert abs(lams[k] - trace[k][1]) <= 1e-8 * trace[k][1]
    np.testing.assert_allclose(p.get_poses(), poses_ref, atol=1e-8)
    assert st.chi2_final < st.chi2_initial


def test_reference_schedule_converges():
    """CGraphG2O::optimizeGraph schedule: 10 x optimize(2) (g2o/g2o_graph.cpp:244-250)."""
    rng = np.random.default_rng(8)
    g = small_graph(rng, n=30, extra=45, noise=0.02)
    p = orc.Problem(**g)
    c0 = p.chi2()
    total = 0
    for _ in range(10):
        rc, st = p.optimize(2)
        assert rc >= 1
        total += rc
    assert total == 20
    assert p.chi2() < 1e-2 * c0 or p.chi2() < 10 * len(g["ei"]) * 6


def test_fixed_vertex_untouched_and_empty_graph():
    rng = np.random.default_rng(9)
    g = small_graph(rng, n=6, extra=4)
    p = orc.Problem(**g)
    p.optimize(3)
    np.testing.assert_array_equal(p.get_poses()[0], g["poses"][0])
    empty = orc.Problem(np.zeros((1, 7)) + [0, 0, 0, 0, 0, 0, 1], [1], [], [], np.zeros((0, 7)), np.zeros((0, 21)))
    rc, _ = empty.optimize(2)
    assert rc == -1                      # g2o: optimize() == -1 when there is nothing to optimise
    assert empty.chi2() == 0.0


def test_openmp_leg_equals_single_thread():
    """The cpu_baseline's one-socket OpenMP leg (bench.py; SURVEY.md §8d ii) is the same arithmetic: the numeric factor is
    bit-identical (rows of independent elimination-tree sub-trees in parallel), the linearisation sums per vertex in the
    same edge order; only the chi2 reduction order differs."""
    import graph_slam_amd as G
    n = 3000
    g = G.synth_manhattan3d(n, 5, 4, seed=21)
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    res = []
    for th in (1, 4):
        orc.set_threads(th)
        try:
            po = orc.Problem(g["poses"], fixed, g["ei"].astype(np.int32), g["ej"].astype(np.int32), g["meas"], g["info"])
            rc, st = po.optimize(3)
            res.append((rc, st.trials, np.array(po.trace()[0]), po.get_poses().copy()))
        finally:
            orc.set_threads(1)
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    np.testing.assert_allclose(res[1][2], res[0][2], rtol=1e-12)
    np.testing.assert_allclose(res[1][3], res[0][3], atol=1e-11)


@pytest.mark.parametrize("threads", [1, 4])
def test_supernodal_leg_equals_simplicial(threads):
    """oracle/orc_chol_sn.c (the second CPU leg of bench.py's cpu_baseline: supernodal left-looking Cholesky on dense panels,
    relaxed amalgamation, OpenMP) against the simplicial restatement of what the reference links: same LM decisions, chi2 to
    1e-12, poses to 1e-10, on a chain-like and on a loop-closure graph (different supernode shapes)"""
    import graph_slam_amd as G
    for n, lookback, loops in ((1500, 4, 0), (4000, 5, 4)):
        g = G.synth_manhattan3d(n, lookback, loops, seed=5)
        fixed = np.zeros(n, np.uint8); fixed[0] = 1
        out = []
        for solver in (0, 1):
            orc.set_solver(solver); orc.set_threads(threads if solver else 1)
            po = orc.Problem(g["poses"], fixed, g["ei"].astype(np.int32), g["ej"].astype(np.int32), g["meas"], g["info"])
            rc = [po.optimize(2)[0] for _ in range(3)]
            out.append((rc, po.chi2(), po.get_poses().copy(), list(po.trace()[0])))
        orc.set_solver(0); orc.set_threads(1)
        assert out[0][0] == out[1][0]
        assert abs(out[0][1] - out[1][1]) <= 1e-12 * out[0][1]
        np.testing.assert_allclose(out[1][3], out[0][3], rtol=1e-11)
        assert np.abs(out[0][2] - out[1][2]).max() < 1e-10
