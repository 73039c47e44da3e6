/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (GTSAM 4.0 un-vendored; the reference's BA experiments only
 * print, gtsam/test/test_ba.cpp:199-202).  GenericProjectionFactor<Pose3, Point3, Cal3DS2> as the reference builds it:
 *   Cal3DS2(fx, fy, 0, cx, cy, k1, k2)                      gtsam/gtsam_graph.cpp:373
 *   GenericProjectionFactor(z, Isotropic::Sigma(2, 1.0), X, Q, K, throwCheirality=false, verbose=false,
 *                           body_P_sensor = *mp_u2c)          gtsam/gtsam_graph.cpp:405-409
 * Semantics restated: camera pose = X * body_P_sensor; p_c = R_c^T (p - t_c);
 * (x, y) = (p_c.x, p_c.y) / p_c.z; radial-tangential distortion; u = fx x_d + s y_d + u0, v = fy y_d + v0;
 * r = (u, v) - z.  Behind the camera with throwCheirality = false: r = 2 fx * (1, 1), Jacobians zero.
 * Jacobians w.r.t. the [omega; v] right perturbation of X and the point; checked against central differences.
 */
#ifndef ORC_CAMERA_H
#define ORC_CAMERA_H
#include "orc_pose3.h"

/* calib[9] = fx fy s u0 v0 k1 k2 p1 p2.  Hx: 2x6 row-major, Hp: 2x3 row-major (either may be NULL). */
static inline void orc_reproj(const double x[7], const double pw[3], const double uv[2], const double calib[9],
                              const double bps[7], double r[2], double *Hx, double *Hp) {
  const double fx = calib[0], fy = calib[1], s = calib[2], u0 = calib[3], v0 = calib[4];
  const double k1 = calib[5], k2 = calib[6], p1 = calib[7], p2 = calib[8];
  double cam[7], Rc[9], d[3], q[3], qc[4];
  orc_pose_mul(x, bps, cam);
  orc_qmat(cam + 3, Rc);
  d[0] = pw[0] - cam[0]; d[1] = pw[1] - cam[1]; d[2] = pw[2] - cam[2];
  orc_qconj(cam + 3, qc);
  orc_qrot(qc, d, q);                                  /* point in the camera frame */
  if (Hx) memset(Hx, 0, 12 * sizeof(double));
  if (Hp) memset(Hp, 0, 6 * sizeof(double));
  if (q[2] <= 0) { r[0] = r[1] = 2.0 * fx; return; }   /* cheirality, throwCheirality = false */
  const double dz = 1.0 / q[2], xn = q[0] * dz, yn = q[1] * dz;
  const double xx = xn * xn, yy = yn * yn, xy = xn * yn, rr = xx + yy;
  const double g = 1. + k1 * rr + k2 * rr * rr;
  const double dx = 2. * p1 * xy + p2 * (rr + 2. * xx), dy = 2. * p2 * xy + p1 * (rr + 2. * yy);
  const double pnx = g * xn + dx, pny = g * yn + dy;
  r[0] = fx * pnx + s * pny + u0 - uv[0];
  r[1] = fy * pny + v0 - uv[1];
  if (!Hx && !Hp) return;
  const double drdx = 2. * xn, drdy = 2. * yn;
  const double dgdx = k1 * drdx + k2 * 2. * rr * drdx, dgdy = k1 * drdy + k2 * 2. * rr * drdy;
  const double dDxdx = 2. * p1 * yn + p2 * (drdx + 4. * xn), dDxdy = 2. * p1 * xn + p2 * drdy;
  const double dDydx = 2. * p2 * yn + p1 * drdx, dDydy = 2. * p2 * xn + p1 * (drdy + 4. * yn);
  const double D00 = g + xn * dgdx + dDxdx, D01 = xn * dgdy + dDxdy, D10 = yn * dgdx + dDydx, D11 = g + yn * dgdy + dDydy;
  /* d(u,v)/d(xn,yn) = [[fx, s], [0, fy]] * D */
  const double A00 = fx * D00 + s * D10, A01 = fx * D01 + s * D11, A10 = fy * D10, A11 = fy * D11;
  /* d(xn,yn)/d(camera pose [omega; v]) (PinholeBase::Dpose) */
  const double Dp[12] = {xy, -(1 + xx), yn, -dz, 0, dz * xn, 1 + yy, -xy, -xn, 0, -dz, dz * yn};
  if (Hx) {
    double Hc[12], binv[7], Ad[36];
    for (int c = 0; c < 6; ++c) { Hc[c] = A00 * Dp[c] + A01 * Dp[6 + c]; Hc[6 + c] = A10 * Dp[c] + A11 * Dp[6 + c]; }
    orc_pose_inv(bps, binv);
    orc_se3_adjoint(binv, Ad);                          /* d compose(X, B) / d X = Ad(B^-1) */
    for (int rr2 = 0; rr2 < 2; ++rr2)
      for (int c = 0; c < 6; ++c) { double t = 0; for (int k = 0; k < 6; ++k) t += Hc[rr2 * 6 + k] * Ad[k * 6 + c]; Hx[rr2 * 6 + c] = t; }
  }
  if (Hp) {
    /* d(xn,yn)/d p_c = [[dz, 0, -dz xn], [0, dz, -dz yn]];  d p_c / d p_w = R_c^T */
    const double E[6] = {dz, 0, -dz * xn, 0, dz, -dz * yn};
    double ER[6];
    for (int a = 0; a < 2; ++a)
      for (int c = 0; c < 3; ++c) ER[a * 3 + c] = E[a * 3] * Rc[c * 3 + 0] + E[a * 3 + 1] * Rc[c * 3 + 1] + E[a * 3 + 2] * Rc[c * 3 + 2];
    for (int c = 0; c < 3; ++c) { Hp[c] = A00 * ER[c] + A01 * ER[3 + c]; Hp[3 + c] = A10 * ER[c] + A11 * ER[3 + c]; }
  }
}
#endif
