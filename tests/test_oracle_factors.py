"""CPU tests of the oracle's remaining GTSAM-semantics factors: reprojection (Cal3DS2 + body_P_sensor) and the
mixed-variable LM (poses + plane landmarks + points, padded 6-blocks).  Parity unpinned for reprojection (no golden
vectors in the reference: gtsam/test/test_ba.cpp only prints); derivative checks pin the restatement."""
import numpy as np
import pytest

from tests import orc_binding as orc
from tests.util import SR4000_CALIB, mixed_graph, mixed_oracle, random_pose, pose_mul, quat_rot, info_full


@pytest.mark.parametrize("seed", range(5))
def test_reprojection_jacobians_vs_central_differences(seed):
    rng = np.random.default_rng(seed)
    x = random_pose(rng)
    bps = random_pose(rng, 0.1)
    calib = SR4000_CALIB.copy()
    if seed >= 3:
        calib[2] = 1.5; calib[7] = 1e-3; calib[8] = -2e-3       # skew + tangential terms too
    cam = pose_mul(x, bps)
    pw = cam[:3] + quat_rot(cam[3:], np.array([rng.uniform(-.5, .5), rng.uniform(-.4, .4), rng.uniform(1.5, 4.0)]))
    uv = rng.normal(size=2) * 3 + [90, 70]
    r, Hx, Hp = orc.reproj(x, pw, uv, calib, bps)
    h = 1e-6
    Nx = np.zeros((2, 6)); Np = np.zeros((2, 3))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        Nx[:, k] = (orc.reproj(orc.retract(x, d), pw, uv, calib, bps, jac=False) - orc.reproj(orc.retract(x, -d), pw, uv, calib, bps, jac=False)) / (2 * h)
    for k in range(3):
        d = np.zeros(3); d[k] = h
        Np[:, k] = (orc.reproj(x, pw + d, uv, calib, bps, jac=False) - orc.reproj(x, pw - d, uv, calib, bps, jac=False)) / (2 * h)
    np.testing.assert_allclose(Hx, Nx, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(Hp, Np, rtol=1e-6, atol=1e-5)


def test_reprojection_cheirality_convention():
    x = np.array([0, 0, 0, 0, 0, 0, 1.0]); bps = x.copy()
    r, Hx, Hp = orc.reproj(x, np.array([0.1, 0.2, -1.0]), np.zeros(2), SR4000_CALIB, bps)
    np.testing.assert_allclose(r, 2 * SR4000_CALIB[0])          # throwCheirality = false: 2 * fx, zero Jacobians
    assert not Hx.any() and not Hp.any()


def test_undistorted_pinhole_known_answer():
    calib = np.array([100.0, 120.0, 0, 50, 40, 0, 0, 0, 0])
    x = np.array([0, 0, 0, 0, 0, 0, 1.0])
    r = orc.reproj(x, np.array([0.5, -0.25, 2.0]), np.zeros(2), calib, x, jac=False)
    np.testing.assert_allclose(r, [100 * 0.25 + 50, 120 * -0.125 + 40])


def test_mixed_graph_dense_system_and_lm():
    rng = np.random.default_rng(3)
    g = mixed_graph(rng)
    p = mixed_oracle(g)
    H, b = p.dense_system()
    assert np.allclose(H, H.T)
    # padded components: identity diagonal, zero rhs, decoupled
    N = len(g["values"])
    for v in range(N):
        if g["vkind"][v] in (orc.VK_PLANE, orc.VK_POINT):
            for r in range(3, 6):
                row = H[6 * v + r].copy(); assert row[6 * v + r] == 1.0
                row[6 * v + r] = 0; assert not row.any() and b[6 * v + r] == 0
    e0 = p.error_gtsam()
    rc, st = p.optimize_gtsam(50)
    assert rc >= 2 and p.error_gtsam() < 0.05 * e0
    vals = p.get_poses()
    for v in range(N):                                   # plane normals stay unit, padding slots untouched
        if g["vkind"][v] == orc.VK_PLANE:
            assert abs(np.linalg.norm(vals[v, :3]) - 1) < 1e-12
