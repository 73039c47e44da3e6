"""BASELINE configs 3 (bundle adjustment) and 4 (visual-inertial + planes) on the GPU, at reduced size and at the full
sizes BASELINE.json names, through size-independent properties:
  * the device's error() against an INDEPENDENT evaluation of all factors (vectorised numpy for the 5M reprojection
    residuals, the oracle's factor functions for the rest),
  * LM drives the error to the noise floor implied by the generator's noise levels,
  * fused chi2 (inside the linearise kernels) == stand-alone chi2 kernels, run-to-run bitwise determinism.
Kernel-level parity of every factor type against the oracle is in test_gpu_gtsam / test_gpu_factors / test_gpu_imu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from graph_slam_amd import scenarios as S
from tests import orc_binding as orc


def _ba_error_numpy(p, poses, points, pose_prior_sigma=1e-3, odo=(0.002, 0.005)):
    """0.5 * sum of squared whitened residuals of the BA graph, evaluated without the product or its kernels"""
    n_kf = len(poses)
    q, t = poses[:, 3:], poses[:, :3]
    bq = np.broadcast_to(p["bps"][3:], (n_kf, 4))
    cam_q = S._quat_mul(q, bq)
    cam_t = t + S._quat_rot(q, np.broadcast_to(p["bps"][:3], (n_kf, 3)))
    kf, pt = p["obs_kf"], p["obs_pt"]
    pk = S._quat_rot(cam_q[kf] * np.array([-1, -1, -1, 1.0]), points[pt] - cam_t[kf])
    r = S._project(pk, p["calib"]) - p["obs_uv"]
    r[pk[:, 2] <= 0] = 2 * p["calib"][0]
    err = 0.5 * np.sum(r * r) / p["pixel_sigma"] ** 2
    err += 0.5 * np.sum((points - p["points0"]) ** 2) / p["point_sigma"] ** 2          # PriorFactor<Point3>
    e0 = orc.prior(poses[0], p["poses"][0], jac=False)
    err += 0.5 * np.sum(e0 * e0) / pose_prior_sigma ** 2
    w = np.array([1 / odo[0] ** 2] * 3 + [1 / odo[1] ** 2] * 3)
    a, b = p["poses"][:-1], p["poses"][1:]
    qa_c = a[:, 3:] * np.array([-1, -1, -1, 1.0])
    meas = np.concatenate([S._quat_rot(qa_c, b[:, :3] - a[:, :3]), S._quat_mul(qa_c, b[:, 3:])], 1)
    for k in range(n_kf - 1):
        e = orc.between(poses[k], poses[k + 1], meas[k], jac=False)
        err += 0.5 * np.sum(w * e * e)
    return err


@pytest.mark.parametrize("n_kf,n_pts", [(300, 8000), (10000, 500000)])
def test_bundle_adjustment(n_kf, n_pts):
    p = S.ba_problem(n_kf, n_pts)
    gr = S.ba_graph(p)
    e0 = gr.error()
    ref0 = _ba_error_numpy(p, p["poses0"], p["points0"])
    assert abs(e0 - ref0) <= 1e-9 * ref0
    rc, st = gr.optimize_gtsam(20)
    assert rc >= 2 and st.n_free == n_kf + n_pts
    vals = gr.get_poses()
    e1 = gr.error()
    ref1 = _ba_error_numpy(p, vals[:n_kf], vals[n_kf:, :3])
    assert abs(e1 - ref1) <= 1e-8 * ref1                       # the optimised state re-evaluated independently
    n_obs = len(p["obs_uv"])
    assert e1 < e0 and 0.8 * n_obs < e1 < 1.2 * n_obs          # 2 residuals / observation at sigma 1 px -> chi2/2 ~ n_obs
    assert abs(st.chi2_final - 2 * e1) <= 1e-9 * e1            # fused chi2 == stand-alone chi2 kernel
    assert np.abs(vals[:n_kf, :3] - p["poses"][:, :3]).max() < 0.02


@pytest.mark.parametrize("n_kf", [400, 50000])
def test_visual_inertial(n_kf):
    p = S.vio_problem(n_kf)
    gr, n_plane_obs = S.vio_graph(p)
    e0 = gr.error()
    rc, st = gr.optimize_gtsam(20)
    e1 = gr.error()
    assert rc >= 2 and st.n_free == 3 * n_kf + len(p["planes"])
    assert e1 < 0.1 * e0
    assert abs(st.chi2_final - 2 * e1) <= 1e-9 * e1
    vals = gr.get_poses(ids=np.arange(n_kf))
    # pulled back onto the truth: the estimate of a chain of noisy relative measurements wanders like sigma * sqrt(n),
    # so the bound is relative to the size of the trajectory (1 %), with a floor for the short case
    extent = np.ptp(p["X"][:, :3], axis=0).max()
    assert np.abs(vals[:, :3] - p["X"][:, :3]).max() < max(0.05, 0.01 * extent)
    planes = gr.get_poses(ids=np.arange(3 * n_kf, 3 * n_kf + len(p["planes"])))
    assert np.abs(np.linalg.norm(planes[:, :3], axis=1) - 1).max() < 1e-12
    # determinism: the same graph again gives bit-identical results
    gr2, _ = S.vio_graph(p)
    gr2.optimize_gtsam(20)
    assert gr2.error() == e1
