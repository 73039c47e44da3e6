"""GenericProjectionFactor<Pose3, Point3, Cal3DS2> with body_P_sensor, a second time -- CHECKER SIDE ONLY (round 6).  From SURVEY.md Appendix A.2's
prose with 4x4 matrices and forward-mode automatic differentiation (tests/se3_independent.Dual), no formula shared with oracle/orc_camera.h or
csrc/factors_device.hpp:  camera pose = X * body_P_sensor;  p_c = (camera pose)^-1 p;  (x, y) = p_c.xy / p_c.z;  r^2 = x^2 + y^2;
g = 1 + k1 r^2 + k2 r^4;  x_d = g x + 2 p1 x y + p2 (r^2 + 2 x^2),  y_d = g y + 2 p2 x y + p1 (r^2 + 2 y^2);  u = fx x_d + s y_d + u0,  v = fy y_d + v0;
residual (u, v) - z;  behind the camera (throwCheirality = false): residual 2 fx (1, 1), Jacobians zero.  The pose moves by the RIGHT perturbation
X Expmap([omega; v]), whose first-order part is X (I + twist matrix): all a first derivative at zero needs.  gtsam/gtsam_graph.cpp:373, 405-409."""
import numpy as np

from tests.se3_independent import Dual, _val, hom, hom_mul, hom_inv, pose_hom


def _twist(d):
    T = np.empty((4, 4), dtype=object)
    w, v = d[:3], d[3:]
    rows = [[1.0, -w[2], w[1], v[0]], [w[2], 1.0, -w[0], v[1]], [-w[1], w[0], 1.0, v[2]], [0.0, 0.0, 0.0, 1.0]]
    for r in range(4):
        for c in range(4):
            T[r, c] = rows[r][c]
    return T


def reproj_ad(x, pw, uv, calib, bps):
    """residual (2), d r / d [omega; v] of the pose (2x6), d r / d point (2x3)"""
    fx, fy, s, u0, v0, k1, k2, p1, p2 = [float(c) for c in calib]
    n = 9
    d = [Dual(0.0, np.eye(n)[k]) for k in range(6)]
    p = [Dual(float(pw[k]), np.eye(n)[6 + k]) for k in range(3)]
    cam = hom_mul(hom_mul(pose_hom(x), _twist(d)), pose_hom(bps))
    ci = hom_inv(cam)                                    # (first order: the inverse of a rotation perturbed by I + [w]x is its transpose)
    pc = [ci[r, 0] * p[0] + ci[r, 1] * p[1] + ci[r, 2] * p[2] + ci[r, 3] for r in range(3)]
    if _val(pc[2]) <= 0:
        return np.array([2 * fx, 2 * fx]), np.zeros((2, 6)), np.zeros((2, 3))
    xn, yn = pc[0] / pc[2], pc[1] / pc[2]
    rr = xn * xn + yn * yn
    g = 1.0 + k1 * rr + k2 * rr * rr
    xd = g * xn + 2.0 * p1 * xn * yn + p2 * (rr + 2.0 * xn * xn)
    yd = g * yn + 2.0 * p2 * xn * yn + p1 * (rr + 2.0 * yn * yn)
    u = fx * xd + s * yd + u0 - float(uv[0])
    v = fy * yd + v0 - float(uv[1])
    r = np.array([_val(u), _val(v)])
    J = np.array([Dual.lift(u, n).g, Dual.lift(v, n).g])
    return r, J[:, :6], J[:, 6:]
