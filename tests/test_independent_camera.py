"""CPU: oracle/orc_camera.h (reprojection residual + closed-form Jacobians) against tests/camera_independent.py (matrices + automatic
differentiation from the prose), SR4000 intrinsics with distortion as the reference sets them (gtsam/test_ba_imu_graph.cpp:84)."""
import numpy as np

from tests import orc_binding as orc
from tests import camera_independent as cam
from tests.util import random_pose, SR4000_CALIB


def test_reprojection_residual_and_jacobians_vs_automatic_differentiation():
    rng = np.random.default_rng(2027)
    behind = 0
    for k in range(400):
        x = random_pose(rng, 1.0)
        bps = random_pose(rng, 0.1)
        calib = SR4000_CALIB.copy()
        if k % 3 == 0:
            calib[7:] = rng.normal(size=2) * 1e-3            # tangential terms too (the reference leaves them zero)
        # a point 1-6 m along the camera's viewing direction with some lateral offset -- and sometimes behind it
        from tests.util import pose_mul, quat_rot
        c = pose_mul(x, bps)
        local = np.array([rng.normal() * 0.4, rng.normal() * 0.4, rng.uniform(1, 6) * (-1 if k % 25 == 0 else 1)])
        pw = c[:3] + quat_rot(c[3:], local)
        uv = rng.uniform(0, 180, size=2)
        r, Hx, Hp = orc.reproj(x, pw, uv, calib, bps)
        r2, Hx2, Hp2 = cam.reproj_ad(x, pw, uv, calib, bps)
        behind += local[2] < 0
        np.testing.assert_allclose(r, r2, atol=1e-9)
        np.testing.assert_allclose(Hx, Hx2, atol=1e-8 * max(1.0, np.abs(Hx2).max()))
        np.testing.assert_allclose(Hp, Hp2, atol=1e-8 * max(1.0, np.abs(Hp2).max()))
    assert behind >= 10                                      # the cheirality branch was part of it
