/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for the GTSAM path: GTSAM 4.0 is not vendored
 * (gtsam/CMakeLists.txt:12-18) and its chart build flags (GTSAM_POSE3_EXPMAP / GTSAM_ROT3_EXPMAP) are unknown,
 * so the full exponential chart is restated and documented:
 *   Pose3 tangent xi = [omega(3); v(3)]  (rotation first: the opposite of g2o)
 *   retract(x, xi) = x * Expmap(xi);  local(x, y) = Logmap(x^-1 y)
 *   PriorFactor<Pose3>    r = Logmap(prior^-1 x),            J = dLog(r)          gtsam_graph.cpp:338-341
 *   BetweenFactor<Pose3>  r = Logmap(Z^-1 xi^-1 xj),         Jj = dLog(r),
 *                                                            Ji = -dLog(r) Ad((xi^-1 xj)^-1)   gtsam_graph.cpp:689-692
 * dLog = Pose3::LogmapDerivative = inverse right Jacobian of SE(3); Ad = Pose3::AdjointMap
 * (used by the reference itself at gtsam_graph.cpp:675-676).  All derivatives are checked against central
 * differences in tests/test_oracle_gtsam.py.  Pose storage as elsewhere: t[3], q[4] = (x,y,z,w).
 */
#ifndef ORC_POSE3_H
#define ORC_POSE3_H
#include "orc_se3.h"

static inline void orc_skew(const double w[3], double S[9]) {
  S[0] = 0; S[1] = -w[2]; S[2] = w[1]; S[3] = w[2]; S[4] = 0; S[5] = -w[0]; S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
static inline void orc_m3mul(const double A[9], const double B[9], double C[9]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
/* Rot3::Expmap as a unit quaternion */
static inline void orc_so3_exp(const double w[3], double q[4]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double s; /* sin(th/2)/th */
  if (th < 1e-10) s = 0.5 - th2 / 48.0; else s = sin(0.5 * th) / th;
  q[0] = s * w[0]; q[1] = s * w[1]; q[2] = s * w[2]; q[3] = cos(0.5 * th);
}
/* Rot3::Logmap from a unit quaternion (angle in [0, pi]) */
static inline void orc_so3_log(const double qin[4], double w[3]) {
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  double k; /* theta / sin(theta/2) */
  if (n < 1e-10) k = 2.0 + n * n / 3.0; else k = 2.0 * atan2(n, q[3]) / n;
  w[0] = k * q[0]; w[1] = k * q[1]; w[2] = k * q[2];
}
/* Rot3::LogmapDerivative: inverse right Jacobian of SO(3) */
static inline void orc_so3_dlog(const double w[3], double J[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double W[9], W2[9];
  orc_skew(w, W);
  orc_m3mul(W, W, W2);
  double c; /* 1/th^2 - (1+cos)/(2 th sin) */
  if (th < 1e-5) c = 1.0 / 12.0 + th2 / 720.0; else c = 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
  for (int k = 0; k < 9; ++k) J[k] = 0.5 * W[k] + c * W2[k];
  J[0] += 1; J[4] += 1; J[8] += 1;
}
/* Pose3::Expmap([w; v]) */
static inline void orc_se3_exp(const double xi[6], double T[7]) {
  const double *w = xi, *v = xi + 3;
  orc_so3_exp(w, T + 3);
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 < 1e-20) { T[0] = v[0]; T[1] = v[1]; T[2] = v[2]; return; }
  const double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
  const double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};   /* w x v */
  double Rc[3];
  orc_qrot(T + 3, c, Rc);
  for (int k = 0; k < 3; ++k) T[k] = (c[k] - Rc[k] + w[k] * wv) / th2;
}
/* Pose3::Logmap -> [w; u] */
static inline void orc_se3_log(const double T[7], double xi[6]) {
  double w[3];
  orc_so3_log(T + 3, w);
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (th < 1e-10) { xi[3] = T[0]; xi[4] = T[1]; xi[5] = T[2]; return; }
  const double a[3] = {w[0] / th, w[1] / th, w[2] / th};
  const double *t = T;
  const double Wt[3] = {a[1] * t[2] - a[2] * t[1], a[2] * t[0] - a[0] * t[2], a[0] * t[1] - a[1] * t[0]};
  const double WWt[3] = {a[1] * Wt[2] - a[2] * Wt[1], a[2] * Wt[0] - a[0] * Wt[2], a[0] * Wt[1] - a[1] * Wt[0]};
  const double k = 1.0 - th / (2.0 * tan(0.5 * th));
  for (int i = 0; i < 3; ++i) xi[3 + i] = t[i] - 0.5 * th * Wt[i] + k * WWt[i];
}
/* Pose3::AdjointMap, row-major 6x6: [[R, 0], [[t]x R, R]] */
static inline void orc_se3_adjoint(const double T[7], double A[36]) {
  double R[9], S[9], SR[9];
  orc_qmat(T + 3, R);
  orc_skew(T, S);
  orc_m3mul(S, R, SR);
  memset(A, 0, 36 * sizeof(double));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { A[r * 6 + c] = R[r * 3 + c]; A[(3 + r) * 6 + 3 + c] = R[r * 3 + c]; A[(3 + r) * 6 + c] = SR[r * 3 + c]; }
}
/* Pose3::LogmapDerivative at xi = Logmap(T): [[Jw, 0], [-Jw Q Jw, Jw]] with Barfoot's Q(xi) */
static inline void orc_se3_dlog(const double xi[6], double J[36]) {
  const double *w = xi, *v = xi + 3;
  double Jw[9], V[9], W[9];
  orc_so3_dlog(w, Jw);
  orc_skew(v, V);
  orc_skew(w, W);
  const double ph2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], ph = sqrt(ph2);
  double WV[9], VW[9], WVW[9], WWV[9], VWW[9], WVWW[9], WWVW[9], WW[9];
  orc_m3mul(W, V, WV); orc_m3mul(V, W, VW); orc_m3mul(WV, W, WVW);
  orc_m3mul(W, W, WW); orc_m3mul(WW, V, WWV); orc_m3mul(VW, W, VWW);
  orc_m3mul(WVW, W, WVWW); orc_m3mul(W, WVW, WWVW);
  double c1, c2, c3;
  if (ph > 1e-5) {
    const double s = sin(ph), c = cos(ph), ph3 = ph2 * ph, ph4 = ph2 * ph2, ph5 = ph4 * ph;
    c1 = (ph - s) / ph3;
    c2 = (1 - ph2 / 2 - c) / ph4;
    c3 = -0.5 * ((1 - ph2 / 2 - c) / ph4 - 3 * (ph - s - ph3 / 6.) / ph5);
  } else {
    c1 = 1. / 6.; c2 = 1. / 24.; c3 = -0.5 * (1. / 24. + 3. / 120.);
  }
  double Q[9], T1[9], Q2[9];
  for (int k = 0; k < 9; ++k)
    Q[k] = -0.5 * V[k] + c1 * (WV[k] + VW[k] - WVW[k]) + c2 * (WWV[k] + VWW[k] - 3 * WVW[k]) + c3 * (WVWW[k] + WWVW[k]);
  orc_m3mul(Jw, Q, T1);
  orc_m3mul(T1, Jw, Q2);
  memset(J, 0, 36 * sizeof(double));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { J[r * 6 + c] = Jw[r * 3 + c]; J[(3 + r) * 6 + 3 + c] = Jw[r * 3 + c]; J[(3 + r) * 6 + c] = -Q2[r * 3 + c]; }
}
/* retract */
static inline void orc_pose3_retract(const double x[7], const double xi[6], double out[7]) {
  double inc[7];
  orc_se3_exp(xi, inc);
  orc_pose_mul(x, inc, out);
  orc_qnormalize(out + 3);
}
static inline void orc_m6mul(const double A[36], const double B[36], double C[36]) {
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += A[r * 6 + k] * B[k * 6 + c]; C[r * 6 + c] = s; }
}
/* BetweenFactor<Pose3> */
static inline void orc_between_pose3(const double xi_[7], const double xj_[7], const double z[7], double e[6], double Ji[36],
                                     double Jj[36]) {
  double xinv[7], h[7], zinv[7], d[7];
  orc_pose_inv(xi_, xinv);
  orc_pose_mul(xinv, xj_, h);            /* h = xi^-1 xj */
  orc_pose_inv(z, zinv);
  orc_pose_mul(zinv, h, d);
  orc_se3_log(d, e);
  if (Ji || Jj) {
    double D[36];
    orc_se3_dlog(e, D);
    if (Jj) memcpy(Jj, D, sizeof(D));
    if (Ji) {
      double hinv[7], A[36];
      orc_pose_inv(h, hinv);
      orc_se3_adjoint(hinv, A);
      orc_m6mul(D, A, Ji);
      for (int k = 0; k < 36; ++k) Ji[k] = -Ji[k];
    }
  }
}
/* PriorFactor<Pose3> */
static inline void orc_prior_pose3(const double x[7], const double prior[7], double e[6], double J[36]) {
  double pinv[7], d[7];
  orc_pose_inv(prior, pinv);
  orc_pose_mul(pinv, x, d);
  orc_se3_log(d, e);
  if (J) orc_se3_dlog(e, J);
}
#endif
