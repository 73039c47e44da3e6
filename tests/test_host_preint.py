"""CPU test of the product's host-side IMU preintegrator (fgo_preint_*, graph_slam_amd/csrc/imu_preint.cpp — the
counterpart of the reference's imu_interface library, gtsam/imu_base.cpp:72-87) against the oracle's restatement:
two independently written implementations of the same recursion must agree to rounding."""
import numpy as np

import graph_slam_amd as G
from tests import orc_binding as orc
from tests.test_oracle_imu import imu_samples


def test_vn100_parameters():
    p = np.zeros(G.IMU_PARAM_DOUBLES)
    G.lib.fgo_imu_params_vn100(G._dp(p))
    d2r = np.pi / 180
    np.testing.assert_allclose(p[0], (0.14e-3 * 9.81) ** 2)                     # imu_vn100.cpp:40,48
    np.testing.assert_allclose(p[1], (0.0035 * d2r) ** 2)
    np.testing.assert_allclose(p[2], 1e-4)
    np.testing.assert_allclose(p[3], ((0.04e-3 * 9.81) * np.sqrt(200)) ** 2)
    np.testing.assert_allclose(p[4], ((10 * d2r / 3600) * np.sqrt(200)) ** 2)
    np.testing.assert_allclose(p[5], 1e-3)
    np.testing.assert_allclose(p[6:], [0, 0, 9.71])                             # MakeSharedD(9.71)


def test_host_preintegrator_matches_oracle():
    rng = np.random.default_rng(0)
    for n in (1, 7, 40, 200):
        acc, gyro = imu_samples(rng, n)
        bhat = rng.normal(size=6) * 0.02
        ref = orc.Preint(bhat, acc, gyro, 0.005)
        pim = G.Preintegrator(bhat)
        for a, w in zip(acc, gyro):
            pim.integrate(a, w, 0.005)
        assert len(pim.buf) == len(ref.buf)                                      # same field layout
        np.testing.assert_allclose(pim.buf[:62], ref.buf[:62], rtol=0, atol=1e-13)            # deltas, Jacobians, bias
        np.testing.assert_allclose(pim.buf[62:], ref.buf[62:], rtol=1e-10, atol=1e-12 * np.abs(ref.buf[62:]).max())   # covariance
        xi = np.concatenate([rng.normal(size=3), [0, 0, 0, 1.0]]); vi = rng.normal(size=3); bi = bhat + 1e-3
        xj, vj = pim.predict(xi, vi, bi)
        xo, vo = ref.predict(xi, vi, bi)
        np.testing.assert_allclose(xj, xo, atol=1e-13); np.testing.assert_allclose(vj, vo, atol=1e-13)


def test_reset_keeps_bias_and_clears_state():
    pim = G.Preintegrator(np.arange(6) * 0.01)
    pim.integrate([0, 0, -9.71], [0.1, 0, 0], 0.005)
    pim.reset(np.ones(6) * 0.5)
    assert pim.buf[0] == 0 and (pim.buf[1:5] == [0, 0, 0, 1]).all() and not pim.buf[62:].any()
    np.testing.assert_array_equal(pim.buf[56:62], 0.5)
